#!/bin/bash
# round 3, batch Z: output intervals queued back to back in rmtExe (integrate_intervals) - tests, then the wall time
mkdir -p gpurun_out/r3z
L=gpurun_out/r3z/log.txt
: > $L
run() { echo "### $*" >> $L; timeout -k 10 900 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $L; }
run python -m pytest tests -m gpu -x -q -k "rmtexe or rmtExe or sweep or ensemble or m2 or outlet or model_setting or member"
run python tools/profile_rmtexe.py profile
run python tools/profile_rmtexe.py outlet
grep -v "^ \|^$" $L | cut -c1-200 | head -40; grep "function calls" $L
