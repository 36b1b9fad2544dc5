set -e
mkdir -p gpurun_out/e4
L=gpurun_out/e4/ktab.txt
: > $L
python tools/run_one.py rk4 dme_nb 1024 256 2000 512 2 reg >> $L 2>&1
python tools/run_one.py rk4 dme_nb 1024 256 2000 512 2 reg RMT_KINETICS_KTAB=1 >> $L 2>&1
python tools/run_one.py rk4 syn12 512 256 500 - - reg >> $L 2>&1
python tools/run_one.py rk4 syn12 512 256 500 - - reg RMT_KINETICS_KTAB=1 >> $L 2>&1
python tools/run_one.py rk45 dme_nb 1024 256 0.008 512 2 auto RMT_RK45_LDS=2 >> $L 2>&1
python tools/run_one.py rk45 dme_nb 1024 256 0.008 512 2 auto RMT_RK45_LDS=2 RMT_KINETICS_KTAB=1 >> $L 2>&1
grep -v "amdgpu.ids\|^accepted" $L
