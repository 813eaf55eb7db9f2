cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_dattn_gpu.py -x -q -m gpu -k "dmlp" 2>&1 | tail -3
b() { python bench.py "$@" --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"; }
for rep in 1 2; do
echo -n "8B S40 dmlpv: "; b --strategy S40
echo -n "8B S40 GBX_DMLP=0: "; GBX_DMLP=0 b --strategy S40
done
echo -n "3B S40 dmlpv: "; b --model llama3.2-3b --strategy S40
echo -n "3B S40 GBX_DMLP=0: "; GBX_DMLP=0 b --model llama3.2-3b --strategy S40
echo -n "8B S22: "; b
echo -n "8B S22 GBX_DMLP=0: "; GBX_DMLP=0 b
