#!/bin/bash
# round 3, batch B: the K-cache (temperature-only rate constants cached per node) in the bench kernel, parity of the
# RK4 paths with it, and the stiff stepper variants on the 12-species mechanism that batch A did not reach.
mkdir -p gpurun_out/r3b
L=gpurun_out/r3b/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" >> $L 2>&1; }
run python -m pytest tests/test_gpu_parity.py -x -q -k "rk4 or geometries or rmtexe_rk4 or full_size_1024"
run python bench.py --no-cpu-baseline --steps 5 --define RMT_KCACHE=0
run python bench.py --no-cpu-baseline --steps 5
run python tools/run_one.py rk4 syn12 512 256 1000 - - auto RMT_KCACHE=0
run python tools/run_one.py rk4 syn12 512 256 1000 - - auto
for g in "256 RMT_ROS_TWOSTEP=1" "256 RMT_ROS_TWOSTEP=0" "128 RMT_ROS_TWOSTEP=0" "64 RMT_ROS_TWOSTEP=0"; do
set -- $g
run timeout -k 10 300 python tools/run_one.py ros4 syn12 512 64 2.0 $1 1 mem $2
done
grep -v "amdgpu.ids" $L | cut -c1-400
