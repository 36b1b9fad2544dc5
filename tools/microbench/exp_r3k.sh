#!/bin/bash
# round 3, batch K: (1) costs of the non-FMA fp64 instructions; (2) ONE 4096-node reactor under the stiff stepper with
# smaller chunks and with the four-lane layout forced for the 7-variable DME mechanism
mkdir -p gpurun_out/r3k
L=gpurun_out/r3k/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 280 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $L; }
run tools/microbench/issue2
run python tools/run_one.py ros4 dme_nb 4096 1 0.5 256 1 chain
run python tools/run_one.py ros4 dme_nb 4096 1 0.5 128 1 chain
run python tools/run_one.py ros4 dme_nb 4096 1 0.5 64 1 chain
run python tools/run_one.py ros4 dme_nb 4096 1 0.5 256 1 chain RMT_ROS_QUAD=1
run python tools/run_one.py ros4 dme_nb 1024 256 0.05 256 1 mem RMT_ROS_QUAD=1
run python tools/run_one.py ros4 dme_nb 1024 256 0.05 256 1 auto RMT_ROS_QUAD=1
run python tools/run_one.py ros4 dme_nb 1024 256 0.05 256 1 auto
cat $L
