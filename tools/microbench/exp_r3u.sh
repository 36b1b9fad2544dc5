#!/bin/bash
# round 3, batch U: padded against unpadded node-major cache rows, twice each
mkdir -p gpurun_out/r3u
L=gpurun_out/r3u/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 400 "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['value'], d['kernel_ms_per_rank'], d['config']['kernel'], d['valu_fp64']['ops_source'][60:160])
    except Exception:
        print(l[:300].rstrip())
" >> $L; }
for i in 1 2; do
run python bench.py --no-cpu-baseline --steps 8
run python bench.py --no-cpu-baseline --steps 8 --define RMT_KC_PAD=0
done
cat $L
