# usage: bash tools/sweep.sh  -- quick A/B of kernel shapes on one GPU
run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.3f G node-steps/s  %.4f ms/step  valu %.3f' % (d['value']/1e9, d['ms_per_step'], d['valu_fp64']['frac']))"; }
run --members 256 --block 512 --npt 2 --lds 2
run --members 512 --block 512 --npt 2 --lds 1
run --members 512 --block 512 --npt 2 --lds 0
run --members 512 --block 256 --npt 4 --lds 1
run --members 1024 --block 256 --npt 4 --lds 0
run --members 1024 --block 512 --npt 2 --lds 0
run --members 2048 --block 512 --npt 2 --lds 2
