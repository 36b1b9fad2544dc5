#!/bin/bash
# Fill the in-tree code-object cache (rmt_app_amd/_kcache/, git-ignored, travels with the gpurun snapshot) with everything
# the GPU test suite JIT-compiles, so that the next `pytest -m gpu` on a fresh box spends its time testing (185 tests:
# 437 s -> 109 s).  Two steps, the first on the GPU box, the second here:
#   gpurun -- 'python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; mkdir -p gpurun_out/kc;
#              tar czf gpurun_out/kc/kcache.tgz -C rmt_app_amd _kcache'
#   tools/harvest_kcache.sh            (unpacks gpurun_out/kc/kcache.tgz into rmt_app_amd/)
# Cache names are digests of source + options + compiler, so an object built from an older template is never loaded;
# after any change to the kernels, the lowering or the prelude the objects are simply rebuilt on first use - harvest again.
set -e
cd "$(dirname "$0")/.."
tar xzf gpurun_out/kc/kcache.tgz -C rmt_app_amd
echo "$(ls rmt_app_amd/_kcache | wc -l) code objects, $(du -sh rmt_app_amd/_kcache | cut -f1)"
