cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dattn_gpu.py -x -q -m gpu 2>&1 | tail -3
one() { python bench.py --strategy S22 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"; }
for rep in 1 2 3 4; do echo -n "MX3=1: "; GBX_DMLP_MX3=1 one; echo -n "  MX3=0: "; GBX_DMLP_MX3=0 one; echo; done
