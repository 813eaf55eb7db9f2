cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py tests/test_model_gpu.py tests/test_generate_fast_gpu.py -x -q -m gpu 2>&1 | tail -4
one() { python bench.py --strategy $1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' | ')"; }
for S in S22 S40; do
  echo -n "$S  MX4=1: "; GBX_DMLP_MX4=1 one $S; GBX_DMLP_MX4=1 one $S; echo; echo -n "$S  MX4=0: "; GBX_DMLP_MX4=0 one $S; GBX_DMLP_MX4=0 one $S; echo
done
