cd $GRAFT_REPO_ROOT
for w in gateup down qkv o; do
  for r in 2 4; do
    echo "--- $w R=$r"
    for o in ${ORDERS:-0 1 2 3}; do
      if [ $o = 0 ]; then unset GBX_QLINEAR_LIB; else export GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_dg_order$o.so; fi
      echo -n "order $o: "; WHICH=$w IMPL=new GBX_DG_R=$r python3 scratch/dg_launch.py 2>/dev/null | tail -1
      echo -n "order $o: "; CARRIED=1 WHICH=$w IMPL=new GBX_DG_R=$r python3 scratch/dg_launch.py 2>/dev/null | tail -1
    done
  done
done
