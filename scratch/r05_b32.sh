cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "chip_wide or slice_folding or batched_runner" 2>&1 | tail -5
run() { echo -n "$1 $2: "; env $1 python bench.py $2 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; }
run GBX_NORM_XCD=1 "--batch 32 --steps 64 --warmup 8"
run GBX_NORM_XCD=0 "--batch 32 --steps 64 --warmup 8"
run GBX_NORM_XCD=1 "--batch 32 --steps 64 --warmup 8"
run GBX_NORM_XCD=0 "--batch 32 --steps 64 --warmup 8"
run GBX_NORM_XCD=1 "--batch 8 --steps 64 --warmup 8"
run GBX_NORM_XCD=0 "--batch 8 --steps 64 --warmup 8"
