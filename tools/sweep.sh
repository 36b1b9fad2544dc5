# usage: bash tools/sweep.sh  -- quick A/B of kernel shapes / tuning macros on one GPU
run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.3f G node-steps/s  %.4f ms/step  %s' % (d['value']/1e9, d['ms_per_step'], d['config']['kernel']))"; }
run
run --nodes 4096 --members 64
run --nodes 4096 --members 256
run --nodes 512 --members 512
