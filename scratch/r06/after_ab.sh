cd $GRAFT_REPO_ROOT
one() { python bench.py --model $1 --strategy $2 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' | ')"; }
for cfg in "llama3.2-3b S40" "llama3-8b S30" "llama3-8b S22"; do
  set -- $cfg
  echo -n "$1 $2  after: "; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_after.so one $1 $2; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_after.so one $1 $2; echo -n "   base: "; one $1 $2; one $1 $2; echo
done
echo -n "llama3-8b S40 MX4=0  after: "; GBX_DMLP_MX4=0 GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_after.so one llama3-8b S40; echo -n "   base: "; GBX_DMLP_MX4=0 one llama3-8b S40; echo
