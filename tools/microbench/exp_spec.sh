set -e
mkdir -p gpurun_out/e5
L=gpurun_out/e5/spec.txt
: > $L
for s in 1 0; do
python tools/run_one.py ros4 dme_nb 1024 256 0.5 256 1 auto SPECIALIZE=$s >> $L 2>&1
python tools/run_one.py rk45 dme_nb 1024 256 0.008 512 2 auto RMT_RK45_LDS=2 SPECIALIZE=$s >> $L 2>&1
python tools/run_one.py rk4 dme_nb 1024 256 2000 512 2 reg SPECIALIZE=$s >> $L 2>&1
done
grep -v "amdgpu.ids" $L
