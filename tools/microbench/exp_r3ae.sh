#!/bin/bash
# round 3, batch AE: -amdgpu-use-amdgpu-trackers=1 on the kernels of the bench line (A/B, twice)
mkdir -p gpurun_out/r3ae
L=gpurun_out/r3ae/log.txt
: > $L
T="COPT=-mllvm -amdgpu-use-amdgpu-trackers=1"
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-250 >> $L; }
for i in 1 2; do
for o in "" "-mllvm -amdgpu-use-amdgpu-trackers=1"; do
echo "### bench copt: $o" >> $L
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 8 --copt "$o" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['kernel_ms_per_rank'])" >> $L
done
run python tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2 auto RMT_RK45_LDS=2
run python tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2 auto RMT_RK45_LDS=2 "$T"
run python tools/run_one.py ros4 dme_nb 1024 256 0.5 256 1 auto
run python tools/run_one.py ros4 dme_nb 1024 256 0.5 256 1 auto "$T"
run python tools/run_one.py rk4 dme_nb 4096 256 300 512 2 chain
run python tools/run_one.py rk4 dme_nb 4096 256 300 512 2 chain "$T"
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain "$T"
done
cat $L
