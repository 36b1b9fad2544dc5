#!/bin/bash
# round 3, batch V: ONE long reactor under the chained RK4 stepper with the chunks' cache in its static form (no second
# code path; the dynamic form did not pay, exp_r3f.sh)
mkdir -p gpurun_out/r3v
L=gpurun_out/r3v/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $L; }
for i in 1 2; do
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain RMT_KCACHE_CHAIN=1 RMT_KCACHE_GEN=0
run python tools/run_one.py rk4 dme_nb 4096 1 4000 128 1 chain RMT_KCACHE_CHAIN=1 RMT_KCACHE_GEN=2 RMT_KC_SMALL_EXP=1 RMT_KC_NODE_MAJOR=1
done
run python tools/run_one.py rk4 dme_nb 16384 1 2000 128 1 chain
run python tools/run_one.py rk4 dme_nb 16384 1 2000 128 1 chain RMT_KCACHE_CHAIN=1 RMT_KCACHE_GEN=2 RMT_KC_SMALL_EXP=1 RMT_KC_NODE_MAJOR=1
cat $L
