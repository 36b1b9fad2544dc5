#!/bin/bash
# round 3, batch P: the chained RK4 stepper with the cache of the temperature-only rate constants (+ redo kernel)
mkdir -p gpurun_out/r3p
L=gpurun_out/r3p/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 500 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-600 >> $L; }
run python -m pytest tests/test_gpu_kcache.py -x -q
for i in 1 2; do
run python tools/run_one.py rk4 dme_nb 4096 256 300 512 2 chain RMT_KCACHE_CHAIN=0
run python tools/run_one.py rk4 dme_nb 4096 256 300 512 2 chain
run python tools/run_one.py rk4 dme_nb 4096 256 300 512 2 chain RMT_KCACHE_CHAIN=0 LDS=1
done
run python tools/run_one.py rk4 dme_nb 16384 64 300 512 2 chain RMT_KCACHE_CHAIN=0
run python tools/run_one.py rk4 dme_nb 16384 64 300 512 2 chain
cat $L
