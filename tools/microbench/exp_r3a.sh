#!/bin/bash
# round 3, batch A: (A) the streaming RHS kernel with / without prefetch and occupancy caps, (B) chunk sizes for ONE
# 4096-node reactor under the chained RK4, (C) what the exp / log of the kinetics cost in the bench kernel and in the
# on-chip RK45 (timing-only switches: results are wrong), (D) the stiff stepper on the 12-species mechanism.
mkdir -p gpurun_out/r3a
L=gpurun_out/r3a/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" >> $L 2>&1; }
run python tools/rhs_stream_bench.py RMT_RHS_PREFETCH=0
run python tools/rhs_stream_bench.py RMT_RHS_PREFETCH=1
run python tools/rhs_stream_bench.py RMT_RHS_PREFETCH=1 RMT_RHS_WAVES=4
for g in "64 1" "128 1" "128 2" "256 1"; do
run python tools/run_one.py rk4 dme_nb 4096 1 4000 $g chain
done
for d in "" "RMT_TIMING_CHEAP_EXP=1" "RMT_TIMING_CHEAP_LOG=1" "RMT_TIMING_CHEAP_EXP=1 RMT_TIMING_CHEAP_LOG=1"; do
run python tools/run_one.py rk4 dme_nb 1024 256 1000 512 2 auto SPECIALIZE=1 $d
done
for d in "" "RMT_TIMING_CHEAP_EXP=1 RMT_TIMING_CHEAP_LOG=1" "RMT_RK45_TWO_COPIES=0"; do
run python tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2 auto SPECIALIZE=1 RMT_RK45_LDS=2 $d
done
for g in "256 RMT_ROS_TWOSTEP=1" "256 RMT_ROS_TWOSTEP=0" "128 RMT_ROS_TWOSTEP=0" "64 RMT_ROS_TWOSTEP=0"; do
set -- $g
run python tools/run_one.py ros4 syn12 512 64 2.0 $1 1 mem $2
done
grep -v "amdgpu.ids" $L
