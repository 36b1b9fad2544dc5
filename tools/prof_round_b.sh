#!/bin/bash
# second half of tools/prof_round.sh: the other steppers, program directly after `--` (tools/prof_cmd.sh)
set -e
tag=${1:-prof}
tools/prof_cmd.sh $tag/ros4_quad_syn12 tools/run_one.py ros4 syn12 512 64 2.0 256 1
tools/prof_cmd.sh $tag/ros4_quad_syn12_mem tools/run_one.py ros4 syn12 512 64 2.0 256 1 mem
tools/prof_cmd.sh $tag/ros4_mem tools/run_one.py ros4 dme_nb 1024 256 0.05 256 1 mem
tools/prof_cmd.sh $tag/ros4_chain tools/run_one.py ros4 dme_nb 4096 1 0.05 256 1 chain
tools/prof_cmd.sh $tag/rk45_reg tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2
tools/prof_cmd.sh $tag/rk4_chain_e1 tools/run_one.py rk4 dme_nb 4096 1 2000
tools/prof_cmd.sh $tag/rk4_chain_e256 tools/run_one.py rk4 dme_nb 4096 256 200
tools/prof_cmd.sh $tag/rhs_stream tools/rhs_stream_one.py
echo "all profiles done"
