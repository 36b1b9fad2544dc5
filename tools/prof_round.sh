#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_round.sh <tag>
# rocprofv3 kernel stats + PMC passes (each in its own run) of the default bench command, the digest-keyed
# HBM traffic record bench.py reads, and the per-kernel profiles of the other steppers -> gpurun_out/<tag>/
set -e
tag=${1:-prof}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline > $out/bench_prof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --no-cpu-baseline > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --no-cpu-baseline > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- python bench.py --no-cpu-baseline > /dev/null
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_misc -- python bench.py --no-cpu-baseline > /dev/null
python tools/summarize_prof.py $out $out/summary > /dev/null
python tools/record_traffic.py $out "profiles/${tag}_bench_rk4_reg.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
cp profiles/traffic.json $out/traffic.json
echo "bench profile done"
# the other steppers: program directly after `--` (tools/prof_cmd.sh)
tools/prof_cmd.sh $tag/ros4_mem tools/run_one.py ros4 dme_nb 1024 256 0.05 256 1 mem
tools/prof_cmd.sh $tag/ros4_chain tools/run_one.py ros4 dme_nb 4096 1 0.05 256 1 chain
tools/prof_cmd.sh $tag/rk45_reg tools/run_one.py rk45 dme_nb 1024 256 8e-3 512 2
tools/prof_cmd.sh $tag/rk45_reg_syn12 tools/run_one.py rk45 syn12 512 64 0.1 256 2
tools/prof_cmd.sh $tag/rk45_chain tools/run_one.py rk45 dme_nb 4096 64 4e-3 512 2 chain RMT_RK45_LDS=2
tools/prof_cmd.sh $tag/rk4_chain_e1 tools/run_one.py rk4 dme_nb 4096 1 2000
tools/prof_cmd.sh $tag/rk4_chain_e256 tools/run_one.py rk4 dme_nb 4096 256 200
tools/prof_cmd.sh $tag/rk4_syn12 tools/run_one.py rk4 syn12 512 256 1000
tools/prof_cmd.sh $tag/ros4_quad_syn12 tools/run_one.py ros4 syn12 512 64 2.0 256 1
tools/prof_cmd.sh $tag/ros4_quad_syn12_mem tools/run_one.py ros4 syn12 512 64 2.0 256 1 mem
echo "all profiles done"
