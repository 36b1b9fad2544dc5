#!/bin/bash
# round 3, batch T: the equilibrium constants cached too (exponent change from basis-function differences, one slot each;
# 64-entry exp table in the caching kernel), node-major padded cache layout
mkdir -p gpurun_out/r3t
L=gpurun_out/r3t/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 400 "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['value'], d['kernel_ms_per_rank'], d['config']['kernel'], d['valu_fp64']['ops_source'][60:330])
    except Exception:
        print(l[:300].rstrip())
" >> $L; }
run python -m pytest tests/test_gpu_kcache.py tests/test_gpu_parity.py -x -q
run python bench.py --no-cpu-baseline --steps 5
run python bench.py --no-cpu-baseline --steps 5
cat $L
