#!/bin/bash
# round 3, batch X: the cache in the one-workgroup stepper at the small geometries (ensembles of short reactors)
mkdir -p gpurun_out/r3x
L=gpurun_out/r3x/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $L; }
C2="RMT_KCACHE=1 RMT_KCACHE_GEN=2 RMT_KC_SMALL_EXP=1 RMT_KC_NODE_MAJOR=1"
C0="RMT_KCACHE=1 RMT_KCACHE_GEN=0"
for shape in "20 2048 64 1" "64 2048 64 1" "100 2048 128 1" "256 1024 256 1" "512 512 512 1" "1024 256 1024 1"; do
set -- $shape
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto RMT_KCACHE=0
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto $C0
run python tools/run_one.py rk4 dme_nb $1 $2 2000 $3 $4 auto $C2 RMT_KC_REFRESH=8
done
cat $L
