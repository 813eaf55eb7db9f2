cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -4
one() { python bench.py --model $1 --strategy $2 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' | ')"; }
for cfg in "llama3.2-3b S40" "llama3-8b S40" "llama3.2-3b S20"; do
  set -- $cfg
  echo -n "$1 $2  MX4=1: "; one $1 $2; one $1 $2; echo -n "   MX4=0: "; GBX_DMLP_MX4=0 one $1 $2; GBX_DMLP_MX4=0 one $1 $2; echo
done
echo -n "llama3.2-3b S20 GBX_DMLP=0: "; GBX_DMLP=0 one llama3.2-3b S20; echo
