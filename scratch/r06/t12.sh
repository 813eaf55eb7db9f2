cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dblock_gpu.py -x -q -m gpu 2>&1 | tail -5
b() { python bench.py "$@" --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"; }
for rep in 1 2 3; do
echo -n "dblock: "; b
echo -n "two-launch: "; GBX_DBLOCK=0 b
done
