cd $GRAFT_REPO_ROOT
for w in gateup qkv o down; do
  for lib in "" $LIBS; do
    if [ -z "$lib" ]; then unset GBX_QLINEAR_LIB; else export GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_$lib.so; fi
    echo -n "${lib:-main}: "; WHICH=$w IMPL=new python3 scratch/dg_launch.py 2>/dev/null | tail -1
  done
done
