# A/B of one environment switch on the B = 32 bench: scratch/ab_env.sh NAME v1 v2 ...
cd $GRAFT_REPO_ROOT
name=$1; shift
for rep in 1 2; do for v in "$@"; do echo -n "$name=$v: "; env $name=$v python bench.py --batch 32 --steps 32 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; done; done
