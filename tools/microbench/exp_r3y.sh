#!/bin/bash
# round 3, batch Y: which cache configuration for the small one-workgroup geometries
mkdir -p gpurun_out/r3y
L=gpurun_out/r3y/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $L; }
C2="RMT_KCACHE=1 RMT_KCACHE_GEN=2 RMT_KC_SMALL_EXP=1 RMT_KC_NODE_MAJOR=1"
C0="RMT_KCACHE=1 RMT_KCACHE_GEN=0"
for shape in "20 2048 64 1" "64 2048 64 1" "100 2048 128 1" "128 2048 128 1" "256 1024 256 1" "200 1024 256 1"; do
set -- $shape
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto RMT_KCACHE=0
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto $C0
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto $C0 RMT_KC_REFRESH=8
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto $C2
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto $C2 RMT_KC_REFRESH=8
done
grep -v "^###" $L | sed 's/rk4 dme_nb //; s/mode=auto //' | cut -c1-230
