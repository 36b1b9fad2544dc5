#!/bin/bash
# round 3, batch O: per-call extremes for the tested stages of the on-chip RK45 stepper; the cached RK4 stepper at
# 1024 x 1 (four waves per SIMD)
mkdir -p gpurun_out/r3o
L=gpurun_out/r3o/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 400 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $L; }
run python tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2 auto RMT_RK45_LDS=2
run python tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2 auto RMT_RK45_LDS=2 RMT_RK45_CALL_FLAGS=1
run python tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2 auto RMT_RK45_LDS=2
run python tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2 auto RMT_RK45_LDS=2 RMT_RK45_CALL_FLAGS=1
run python bench.py --no-cpu-baseline --steps 5 --block 1024 --npt 1 --lds 1 --define RMT_KCACHE=1 --define RMT_KCACHE_GEN=0
run python bench.py --no-cpu-baseline --steps 5 --block 1024 --npt 1 --lds 2
cat $L
