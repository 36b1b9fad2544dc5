set -e
mkdir -p gpurun_out/e6
L=gpurun_out/e6/sched.txt
: > $L
for o in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=iterative-minreg"; do
python tools/run_one.py ros4 dme_nb 1024 256 0.5 256 1 auto "COPT=-mllvm -disable-machine-licm $o" >> $L 2>&1
done
grep -v "amdgpu.ids\|^accepted" $L
