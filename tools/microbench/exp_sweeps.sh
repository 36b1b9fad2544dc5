set -e
mkdir -p gpurun_out/e3
L=gpurun_out/e3/ktab.txt
: > $L
python tools/run_one.py ros4 dme_nb 1024 256 0.5 256 1 auto RMT_KINETICS_KTAB=1 >> $L 2>&1
python tools/run_one.py ros4 syn12 512 64 2.0 - - auto RMT_KINETICS_KTAB=1 >> $L 2>&1
grep -v "amdgpu.ids" $L
