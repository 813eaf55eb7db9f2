cd $GRAFT_REPO_ROOT
one() { python bench.py --context $1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"; }
for C in 300 400 500 600 800 1000 1400; do
  echo -n "C=$C  one launch: "; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_ctx2k.so one $C; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_ctx2k.so one $C
  echo -n "  two launches: "; GBX_DBLOCK=0 one $C; GBX_DBLOCK=0 one $C; echo
done
