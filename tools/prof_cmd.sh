#!/bin/bash
# usage (GPU box, repo root): tools/prof_cmd.sh <tag> <python-script> [args...]
# rocprofv3 kernel stats + SQ / LDS-VMEM / I-cache PMC passes (each in its own run) of
#   python <script> [args]   ->  gpurun_out/<tag>/summary.md   (copy what should be judged into profiles/)
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python "$@" > $out/run.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- python "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $out/pmc_misc -- python "$@" > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $out/pmc_ic -- python "$@" > /dev/null 2>&1
python tools/summarize_prof.py $out $out/summary > /dev/null
echo "profile written to gpurun_out/$tag/summary.md"
