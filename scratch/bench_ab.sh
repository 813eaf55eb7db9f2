# A/B of dgemv switches on the REAL token graph (bench.py headline, 128 steps)
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python bench.py --steps 128 --warmup 8 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; }
for v in ${VARIANTS}; do run "$v"; done
