#!/bin/bash
# round 3, batch C: K-cache variants of the bench kernel (stage-level mode selection)
mkdir -p gpurun_out/r3c
L=gpurun_out/r3c/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" >> $L 2>&1; }
run python bench.py --no-cpu-baseline --steps 5 --define RMT_KCACHE=0
run python bench.py --no-cpu-baseline --steps 5
run python bench.py --no-cpu-baseline --steps 5 --define RMT_STAGE_UNROLL=0
run python bench.py --no-cpu-baseline --steps 5 --define RMT_STAGE_UNROLL=0 --define RMT_KCACHE=0
run python -m pytest tests/test_gpu_parity.py -x -q -k "rk4 or geometries or rmtexe_rk4 or full_size_1024"
grep -v "amdgpu.ids" $L | cut -c1-330
