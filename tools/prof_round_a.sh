#!/bin/bash
# first half of tools/prof_round.sh (fits one gpurun call): rocprofv3 kernel stats + PMC passes of the default bench
# command and the digest-keyed HBM traffic record -> gpurun_out/<tag>/
set -e
tag=${1:-prof}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_default.json
echo "bench line written"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline > $out/bench_prof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --no-cpu-baseline > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --no-cpu-baseline > /dev/null
echo "traffic passes done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- python bench.py --no-cpu-baseline > /dev/null
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_misc -- python bench.py --no-cpu-baseline > /dev/null
python tools/summarize_prof.py $out $out/summary > /dev/null
python tools/record_traffic.py $out "profiles/${tag}_bench_rk4_reg.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
cp profiles/traffic.json $out/traffic.json
echo "bench profile done"
