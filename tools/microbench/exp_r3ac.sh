#!/bin/bash
# round 3, batch AC: model M2 through the caching on-chip RK4 stepper
mkdir -p gpurun_out/r3ac
L=gpurun_out/r3ac/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 400 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $L; }
run python -m pytest tests/test_gpu_kcache.py tests/test_gpu_m2.py -x -q




run python tools/microbench/m2_cache_rate.py
grep -v "^###" $L | cut -c1-220 | tail -8
