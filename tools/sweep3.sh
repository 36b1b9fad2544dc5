run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.3f G node-steps/s  %.4f ms/step' % (d['value']/1e9, d['ms_per_step']))"; }
run
run --copt "-mllvm -amdgpu-sched-strategy=max-ilp"
run --copt "-mllvm -amdgpu-sched-strategy=max-memory-clause"
run --block 1024 --npt 1 --copt "-mllvm -amdgpu-sched-strategy=max-ilp"
run --copt "-mllvm -disable-machine-licm"
