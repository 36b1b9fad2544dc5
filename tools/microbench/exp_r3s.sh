#!/bin/bash
# round 3, batch S: (timing only, wrong results) what caching the three equilibrium constants with polynomial exponents
# (historical: the stand-in switch RMT_TIMING_GEN_CHEAP was removed from the lowering after this measurement)
# as well would be worth - their exp replaced by a two-instruction stand-in - and the merged node reciprocals
mkdir -p gpurun_out/r3s
L=gpurun_out/r3s/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['value'], d['kernel_ms_per_rank'], d['config']['kernel'], d['valu_fp64']['ops_source'][:160])
    except Exception:
        print(l[:300].rstrip())
" >> $L; }
run python bench.py --no-cpu-baseline --steps 5 --define RMT_NODE_RCP_MERGE=0
run python bench.py --no-cpu-baseline --steps 5
export RMT_TIMING_GEN_CHEAP=1
run python bench.py --no-cpu-baseline --steps 5
run python tools/run_one.py rk4 dme_nb 1024 256 1000 512 2 auto
cat $L
