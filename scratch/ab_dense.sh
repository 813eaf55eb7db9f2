cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fused_gpu.py -q -x -k "dense_forward" 2>&1 | tail -2
for v in 1 0 1 0; do echo -n "GBX_DENSE_SKINNY=$v: "; GBX_DENSE_SKINNY=$v python bench.py --batch 32 --steps 32 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; done
