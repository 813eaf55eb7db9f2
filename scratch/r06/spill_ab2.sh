cd $GRAFT_REPO_ROOT
one() { python bench.py --strategy S30 --context $1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"; }
for C in 64 100 128 140 200 300; do
  echo -n "S30 C=$C  new: "; one $C; one $C; echo -n "  old: "; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so one $C; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so one $C; echo
done
