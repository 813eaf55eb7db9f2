cd $GRAFT_REPO_ROOT
for n in "$@"; do
  lib=""; [ "$n" != default ] && lib=$PWD/scratch/exp/lib_$n.so
  echo "== $n"; GBX_QLINEAR_LIB=$lib python scratch/bench_skinny32.py 2>&1 | tail -4
done
