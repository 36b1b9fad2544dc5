#!/bin/bash
# round 3, batch AF: a few more LLVM switches on the final bench kernel (on top of the trackers)
mkdir -p gpurun_out/r3af
L=gpurun_out/r3af/log.txt
: > $L
for o in "" "-mllvm -enable-post-misched=0" "-mllvm -amdgpu-schedule-relaxed-occupancy=1" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -amdgpu-schedule-metric-bias=30" "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1" "-mllvm -amdgpu-enable-rewrite-partial-reg-uses=1" "-mllvm -misched-cluster=0"; do
echo "### copt: $o" >> $L
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --copt "$o" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['value'], d['kernel_ms_per_rank'], d['valu_fp64']['ops_source'][60:150])
    except Exception:
        if 'rror' in l or 'nknown' in l: print(l[:160].rstrip())
" >> $L
done
cat $L
