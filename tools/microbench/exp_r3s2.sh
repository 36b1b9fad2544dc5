#!/bin/bash
# batch S, second half: the same under rocprofv3 --kernel-trace --stats, to read the duration of rmt_n2_rk4_reg alone
# (historical: the stand-in switch RMT_TIMING_GEN_CHEAP was removed from the lowering after this measurement)
# (with the stand-in the reactors leave the cache's range and the redo kernel integrates them again)
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3s
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/base -- python bench.py --no-cpu-baseline --steps 5 > $out/base.log 2>&1
export RMT_TIMING_GEN_CHEAP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/cheap -- python bench.py --no-cpu-baseline --steps 5 > $out/cheap.log 2>&1
for v in base cheap; do echo "## $v"; cat $out/$v/*/*kernel_stats.csv | cut -c1-200; done
