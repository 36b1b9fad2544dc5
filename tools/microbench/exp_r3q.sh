#!/bin/bash
# round 3, batch Q: what the one workgroup barrier per RHS evaluation costs the cached bench kernel (timing only:
# RMT_TIMING_NO_BARRIER drops it, the results are wrong)
mkdir -p gpurun_out/r3q
L=gpurun_out/r3q/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $L; }
for i in 1 2; do
run python tools/run_one.py rk4 dme_nb 1024 256 1000 512 2 auto
run python tools/run_one.py rk4 dme_nb 1024 256 1000 512 2 auto RMT_TIMING_NO_BARRIER=1
done
cat $L
