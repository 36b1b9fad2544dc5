#!/bin/bash
# round 3, batch H: two-step sweeps in the quad layout (12-species stiff job) + the tests touched since the last full run
mkdir -p gpurun_out/r3h
L=gpurun_out/r3h/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" 2>&1 | cut -c1-400 >> $L; }
run timeout -k 10 200 python tools/run_one.py ros4 syn12 512 64 2.0 256 1 auto
run timeout -k 10 200 python tools/run_one.py ros4 syn12 512 64 2.0 256 1 mem
run timeout -k 10 200 python tools/run_one.py ros4 syn12 512 64 2.0 256 1 auto RMT_ROSQ_TWOSTEP=0
run timeout -k 10 400 python -m pytest tests/test_gpu_config5.py tests/test_gpu_parity.py -x -q -k "quad or syn12 or kcache"
grep -v "amdgpu.ids" $L | cut -c1-300
