#!/bin/bash
# round 3, batch R: the cache's reference point moved every K steps instead of every step (stage 1 of the steps in
# between takes the cached path too)
mkdir -p gpurun_out/r3r
L=gpurun_out/r3r/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['value'], d['kernel_ms_per_rank'], d['config']['kernel'], d['valu_fp64']['ops_source'])
    except Exception:
        print(l[:300].rstrip())
" >> $L; }
for k in 1 4 8 16; do
run python bench.py --no-cpu-baseline --steps 5 --define RMT_KC_REFRESH=$k
done
cat $L
