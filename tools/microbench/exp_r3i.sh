#!/bin/bash
# round 3, batch I: the streaming RHS kernel with workgroups persistent over the reactors (cap per CU swept)
mkdir -p gpurun_out/r3i
L=gpurun_out/r3i/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" 2>&1 | cut -c1-400 >> $L; }
run timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_plain.py tests/test_gpu_config5.py -x -q -k "rhs or 16384 or cabi"
for w in 1000000 16 8 4 2; do
  echo "### RMT_N2_RHS_WGS_PER_CU=$w" >> $L
  RMT_N2_RHS_WGS_PER_CU=$w timeout -k 10 200 python tools/rhs_stream_bench.py 2>&1 | cut -c1-400 >> $L
done
grep -v "amdgpu.ids" $L | cut -c1-300
