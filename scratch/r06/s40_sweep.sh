cd $GRAFT_REPO_ROOT
one() { python bench.py --strategy S40 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"; }
for v in "X=1" "GBX_DG_DEPTH=3" "GBX_DG_DEPTH=4" "GBX_DG_LATE=0" "GBX_DG_P2_NW=4" "GBX_DG_P2_NW=16" "GBX_DG_R=4" "GBX_DG_MIN_NW=8" "X=1"; do
  echo -n "$v: "; env $v bash -c "$(declare -f one); one; one"; echo
done
