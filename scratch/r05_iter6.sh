cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5
run() { echo -n "$1 $2: "; env $1 python bench.py $2 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; }
run GBX_DATTN=1 "--steps 20 --warmup 8"
run GBX_DATTN_COOP_MIN=100000 "--warmup 8"
run GBX_DATTN_COOP_MIN=160 "--warmup 8"
run GBX_DATTN_COOP_MIN=100 "--warmup 8"
run GBX_DATTN_COOP_MIN=100000 "--warmup 8 --context 200 --steps 20"
run GBX_DATTN_COOP_MIN=160 "--warmup 8 --context 200 --steps 20"
run GBX_DATTN_COOP_MIN=100000 "--warmup 8 --context 300 --steps 20"
run GBX_DATTN_COOP_MIN=160 "--warmup 8 --context 300 --steps 20"
run GBX_DATTN_COOP_MIN=100000 "--warmup 8 --context 600 --steps 20"
run GBX_DATTN_COOP_MIN=160 "--warmup 8 --context 600 --steps 20"
run GBX_DATTN=1 "--warmup 8 --context 2000 --steps 20"
run GBX_DATTN=1 "--warmup 8 --context 6000 --steps 20"
run GBX_DATTN=1 "--warmup 8 --context 16000 --steps 20"
run GBX_DATTN=0 "--warmup 8 --context 16000 --steps 20"
