cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_$1.so python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"; }
for rep in 1 2; do for v in "$@"; do run $v; done; done
echo -n "two-launch: "; GBX_DBLOCK=0 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"
