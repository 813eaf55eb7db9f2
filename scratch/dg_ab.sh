cd $GRAFT_REPO_ROOT
for w in gateup qkv o down; do
  for lib in "" $LIBS; do
    if [ -z "$lib" ]; then unset GBX_QLINEAR_LIB; else export GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_$lib.so; fi
    echo -n "${lib:-main}: "; WHICH=$w IMPL=new python3 scratch/dg_launch.py 2>&1 | tail -1
    if [ -n "$RS" ]; then for r in $RS; do echo -n "${lib:-main}: "; GBX_DG_R=$r WHICH=$w IMPL=new python3 scratch/dg_launch.py 2>&1 | tail -1; done; fi
  done
done
