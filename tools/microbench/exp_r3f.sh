#!/bin/bash
# round 3, batch F: K-cache in the chained RK4 stepper (one long reactor, one node per lane)
mkdir -p gpurun_out/r3f
L=gpurun_out/r3f/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" >> $L 2>&1; }
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain RMT_KCACHE_CHAIN=1
run python tools/run_one.py rk4 dme_nb 16384 1 2000 128 1 chain
run python tools/run_one.py rk4 dme_nb 16384 1 2000 128 1 chain RMT_KCACHE_CHAIN=1
run python tools/run_one.py rk4 dme_nb 1024 64 1000 128 1 chain
run python tools/run_one.py rk4 dme_nb 1024 64 1000 128 1 chain RMT_KCACHE_CHAIN=1
run python tools/run_one.py rk4 dme_nb 1024 64 1000 256 1 chain RMT_KCACHE_CHAIN=1
grep -v "amdgpu.ids" $L | cut -c1-330
