#!/bin/bash
# round 3, batch AD: scheduler options on the final bench kernel
mkdir -p gpurun_out/r3ad
L=gpurun_out/r3ad/log.txt
: > $L
for o in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=iterative-minreg" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-use-amdgpu-trackers=1"; do
echo "### copt: $o" >> $L
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --copt "$o" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['value'], d['kernel_ms_per_rank'], d['valu_fp64']['ops_source'][60:150])
    except Exception:
        print(l[:200].rstrip())
" >> $L
done
cat $L
