#!/bin/bash
# round 3, batch L: the K-cache without a second code path (stage 1 in full, stages 2-4 from the cache; timing only - the
# range test is off) - the upper bound of what caching the temperature-only constants can give the bench kernel
mkdir -p gpurun_out/r3l
L=gpurun_out/r3l/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 280 "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['value'], d['kernel_ms_per_rank'], d['config']['kernel'], d['valu_fp64']['ops_source'])
    except Exception:
        print(l[:300].rstrip())
" >> $L; }
run python bench.py --no-cpu-baseline --steps 5
run python bench.py --no-cpu-baseline --steps 5 --lds 1
run python bench.py --no-cpu-baseline --steps 5 --lds 0 --define RMT_KCACHE=1 --define RMT_KCACHE_STATIC=1 --define RMT_KCACHE_GEN=0
run python bench.py --no-cpu-baseline --steps 5 --lds 1 --define RMT_KCACHE=1 --define RMT_KCACHE_STATIC=1 --define RMT_KCACHE_GEN=0
cat $L
