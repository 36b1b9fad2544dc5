#!/bin/bash
# round 3, batch W: chained cache at the other chained geometries, and with the reference point moved every 8th step
mkdir -p gpurun_out/r3w
L=gpurun_out/r3w/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $L; }
C="RMT_KCACHE_CHAIN=1 RMT_KCACHE_GEN=0"
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain $C
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain $C RMT_KC_REFRESH_CHAIN=8
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain $C RMT_KC_REFRESH_CHAIN=8
run python tools/run_one.py rk4 dme_nb 1024 64 2000 256 1 chain RMT_KCACHE_CHAIN=0
run python tools/run_one.py rk4 dme_nb 1024 64 2000 256 1 chain $C
run python tools/run_one.py rk4 dme_nb 1024 64 2000 256 1 chain $C RMT_KC_REFRESH_CHAIN=8
run python tools/run_one.py rk4 dme_nb 1024 128 2000 256 2 chain RMT_KCACHE_CHAIN=0
run python tools/run_one.py rk4 dme_nb 1024 128 2000 256 2 chain $C
run python tools/run_one.py rk4 dme_nb 1024 128 2000 256 2 chain $C RMT_KC_REFRESH_CHAIN=8
run python tools/run_one.py rk4 dme_nb 1024 32 2000 128 1 chain RMT_KCACHE_CHAIN=0
run python tools/run_one.py rk4 dme_nb 1024 32 2000 128 1 chain $C RMT_KC_REFRESH_CHAIN=8
cat $L
