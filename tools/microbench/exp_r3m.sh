#!/bin/bash
# round 3, batch M: the cached on-chip RK4 stepper with its redo kernel - parity, then the bench line
mkdir -p gpurun_out/r3m
L=gpurun_out/r3m/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 500 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-3000 >> $L; }
run python -m pytest tests/test_gpu_kcache.py -x -q
run python bench.py --no-cpu-baseline --steps 5
run python bench.py --no-cpu-baseline --steps 5
cut -c1-400 $L
