run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.3f G node-steps/s  %.4f ms/step' % (d['value']/1e9, d['ms_per_step']))"; }
run --block 512 --npt 2 --lds 2
run --block 1024 --npt 1 --lds 2
run --block 1024 --npt 1 --lds 2 --define RMT_STAGE_UNROLL=0
run --block 512 --npt 2 --lds 2 --define RMT_STAGE_UNROLL=0
run --block 512 --npt 1 --lds 2 --nodes 512 --members 512
run --block 256 --npt 2 --lds 2 --nodes 512 --members 512
