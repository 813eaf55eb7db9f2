cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -8
one() { python bench.py --model $1 --strategy $2 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config'].get('block_launches','')[:60], end=' | ')"; }
for cfg in "llama3-8b S22" "llama3-8b S30" "llama3.2-3b S30"; do
  set -- $cfg
  echo -n "$1 $2  MX3=1: "; one $1 $2; one $1 $2; echo; echo -n "$1 $2  MX3=0: "; GBX_DMLP_MX3=0 one $1 $2; GBX_DMLP_MX3=0 one $1 $2; echo
done
