#!/bin/bash
# round 3, batch AB: DME at 512 x 1 with the cache AND y_n in LDS (the default lds_state measured slower, batch X)
mkdir -p gpurun_out/r3ab
L=gpurun_out/r3ab/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $L; }
C2="RMT_KCACHE=1 RMT_KCACHE_GEN=2 RMT_KC_SMALL_EXP=1 RMT_KC_NODE_MAJOR=1 RMT_KC_REFRESH=8"
C0="RMT_KCACHE=1 RMT_KCACHE_GEN=0 RMT_KC_REFRESH=8"
for n in 512 400; do
run python tools/run_one.py rk4 dme_nb $n 512 2000 512 1 auto RMT_KCACHE=0
run python tools/run_one.py rk4 dme_nb $n 512 2000 512 1 auto $C0 LDS=1
run python tools/run_one.py rk4 dme_nb $n 512 2000 512 1 auto $C2 LDS=1
run python tools/run_one.py rk4 dme_nb $n 512 2000 512 1 auto $C2 LDS=0
done
grep -v "^###" $L | sed 's/rk4 dme_nb //; s/mode=auto //' | cut -c1-230
