cd $GRAFT_REPO_ROOT
for v in 1 0; do
  echo "== GBX_BATCH_V2=$v"
  GBX_BATCH_V2=$v BITS=${BITS:-2} python scratch/bench_skinny32.py 2>&1 | tail -n +2
  GBX_BATCH_V2=$v python bench.py --batch 32 --steps 32 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"
done
