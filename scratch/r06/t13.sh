cd $GRAFT_REPO_ROOT
b() { python bench.py "$@" --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"; }
for c in 200 300 350; do
echo -n "ctx $c dblock: "; b --context $c
echo -n "ctx $c two-launch: "; GBX_DBLOCK=0 b --context $c
done
echo -n "512 steps dblock: "; python bench.py --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"
echo -n "512 steps two-launch: "; GBX_DBLOCK=0 python bench.py --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"
