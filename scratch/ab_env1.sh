# A/B of one environment switch on the batch-1 bench (driver-style and 128 steps): scratch/ab_env1.sh NAME v1 v2 ...
cd $GRAFT_REPO_ROOT
name=$1; shift
for rep in 1 2; do for v in "$@"; do for st in 20 128; do echo -n "$name=$v steps=$st: "; env $name=$v python bench.py --gpus 1 --steps $st --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; done; done; done
