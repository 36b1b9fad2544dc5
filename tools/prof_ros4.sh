#!/bin/bash
# usage (GPU box, repo root): tools/prof_ros4.sh <tag>  - rocprofv3 stats + HBM PMC passes of the stiff stepper
# on the bench ensemble (256 x 1024, 0.05 s of the transient, default tolerances)
set -e
tag=${1:-ros4}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python tools/tts_bench.py 256 1024 0.05 SKIP_RK4=1 > $out/tts.md
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python tools/tts_bench.py 256 1024 0.05 SKIP_RK4=1 > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python tools/tts_bench.py 256 1024 0.05 SKIP_RK4=1 > /dev/null
python tools/summarize_prof.py $out $out/summary
