#!/bin/bash
set -e
tag=${1:-ros4sq}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- python tools/tts_bench.py 256 1024 0.05 SKIP_RK4=1 > /dev/null
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $out/pmc_misc -- python tools/tts_bench.py 256 1024 0.05 SKIP_RK4=1 > /dev/null
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $out/pmc_ic -- python tools/tts_bench.py 256 1024 0.05 SKIP_RK4=1 > /dev/null
python tools/summarize_prof.py $out $out/summary
