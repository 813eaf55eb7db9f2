# A/B of the batch GEMM on one box: default library vs scratch/exp/lib_$1.so
cd $GRAFT_REPO_ROOT
for lib in "" "$PWD/scratch/exp/lib_$1.so"; do
  echo "== lib: ${lib:-default}"
  GBX_QLINEAR_LIB=$lib python scratch/bench_skinny32.py 2>&1 | tail -4
  GBX_QLINEAR_LIB=$lib python bench.py --batch 32 --steps 32 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"
done
