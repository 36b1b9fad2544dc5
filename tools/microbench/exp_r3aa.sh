#!/bin/bash
# round 3, batch AA: the cache on the 12-species mechanism (V = 13, eight Arrhenius constants)
mkdir -p gpurun_out/r3aa
L=gpurun_out/r3aa/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $L; }
run python tools/run_one.py rk4 syn12 512 256 1000 - - auto
run python tools/run_one.py rk4 syn12 512 256 1000 - - auto RMT_KCACHE=1 RMT_KCACHE_GEN=0 RMT_KC_REFRESH=8 LDS=1
run python tools/run_one.py rk4 syn12 512 256 1000 - - auto RMT_KCACHE=1 RMT_KCACHE_GEN=0 RMT_KC_REFRESH=8 RMT_KC_NODE_MAJOR=1 LDS=1
run python tools/run_one.py rk4 syn12 1024 256 500 - - auto
run python tools/run_one.py rk4 syn12 1024 256 500 - - auto RMT_KCACHE_CHAIN=1 RMT_KCACHE_GEN=0 LDS=1
run python tools/run_one.py rk4 syn12 128 2048 1000 - - auto
run python tools/run_one.py rk4 syn12 128 2048 1000 - - auto RMT_KCACHE=1 RMT_KCACHE_GEN=0 RMT_KC_REFRESH=8
run python tools/run_one.py rk4 syn12 64 2048 1000 - - auto
run python tools/run_one.py rk4 syn12 64 2048 1000 - - auto RMT_KCACHE=1 RMT_KCACHE_GEN=0 RMT_KC_REFRESH=8
grep -v "^###" $L | sed 's/rk4 syn12 //; s/mode=auto //' | cut -c1-220
