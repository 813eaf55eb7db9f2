# A/B of an experiment library against the in-tree one on the batch-1 bench: scratch/ab_lib1.sh <name>
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" "$PWD/scratch/exp/lib_$1.so"; do for st in 20 128; do echo -n "lib=${lib:-default} steps=$st: "; GBX_QLINEAR_LIB=$lib python bench.py --gpus 1 --steps $st --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; done; done; done
