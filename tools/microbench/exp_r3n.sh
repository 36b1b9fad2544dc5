#!/bin/bash
# round 3, batch N: stage-1 exception tests as per-call extremes (RMT_CALL_FLAGS) in the cached on-chip RK4 stepper
mkdir -p gpurun_out/r3n
L=gpurun_out/r3n/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 500 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-3000 >> $L; }
run python bench.py --no-cpu-baseline --steps 5
run python bench.py --no-cpu-baseline --steps 5 --define RMT_CALL_FLAGS=1
cut -c1-300 $L
