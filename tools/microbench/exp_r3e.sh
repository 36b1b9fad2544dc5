#!/bin/bash
# round 3, batch E: the quad-layout stiff stepper (V > 8): parity first, then the 64 x 512 x 2 s job
mkdir -p gpurun_out/r3e
L=gpurun_out/r3e/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" >> $L 2>&1; }
run timeout -k 10 400 python -m pytest tests/test_gpu_config5.py -x -q -k "quad"
run timeout -k 10 120 python tools/run_one.py ros4 syn12 512 64 2.0 256 1 mem
run timeout -k 10 120 python tools/run_one.py ros4 syn12 512 64 2.0 256 1 auto
run timeout -k 10 120 python tools/run_one.py ros4 syn12 512 64 2.0 256 1 chain
run timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config5.py -x -q -k "syn12 or ros4"
grep -v "amdgpu.ids" $L | cut -c1-330
