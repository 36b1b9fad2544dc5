#!/bin/bash
# round 3, batch G: bench kernel geometries with more nodes per lane (one wave per SIMD, 512 registers)
mkdir -p gpurun_out/r3g
L=gpurun_out/r3g/log.txt
: > $L
run() { echo "### $*" >> $L; "$@" 2>&1 | cut -c1-400 >> $L; }
run python bench.py --no-cpu-baseline --steps 5 --block 256 --npt 4
run python bench.py --no-cpu-baseline --steps 5 --block 256 --npt 4 --lds 1
run python bench.py --no-cpu-baseline --steps 5 --block 256 --npt 4 --lds 0
run python bench.py --no-cpu-baseline --steps 5 --block 512 --npt 2 --lds 2
grep -v "amdgpu.ids" $L | cut -c1-300
