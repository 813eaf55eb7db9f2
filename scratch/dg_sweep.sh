# old vs new decode launches, geometry sweep (scratch/dg_launch.py)
cd $GRAFT_REPO_ROOT
for w in qkv o gateup down; do
  WHICH=$w IMPL=old python3 scratch/dg_launch.py 2>/dev/null | tail -1
  for r in 2 4; do for nw in 4 8 16; do
    WHICH=$w IMPL=new GBX_DG_R=$r GBX_DG_MIN_NW=$nw python3 scratch/dg_launch.py 2>/dev/null | tail -1
  done; done
  WHICH=$w IMPL=new CARRIED=1 python3 scratch/dg_launch.py 2>/dev/null | tail -1
done
