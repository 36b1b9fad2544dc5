set -e
mkdir -p gpurun_out/e1
L=gpurun_out/e1/log2.txt
: > $L
M="-mllvm -disable-machine-licm"
R="-mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=iterative-minreg"
python tools/run_one.py rk45 dme_nb 1024 256 0.008 512 2 "COPT=$M" >> $L 2>&1
python tools/run_one.py rk45 dme_nb 1024 256 0.008 512 2 "COPT=$R" >> $L 2>&1
python tools/run_one.py rk45 syn12 512 64 0.1 256 2 "COPT=$R" >> $L 2>&1
python tools/run_one.py ros4 dme_nb 1024 256 0.05 256 1 "COPT=$R" >> $L 2>&1
python tools/run_one.py rk45 dme_nb 1024 256 0.008 512 2 "COPT=$M -mllvm -disable-machine-sink" >> $L 2>&1
grep -v "^accepted\|amdgpu.ids" $L
