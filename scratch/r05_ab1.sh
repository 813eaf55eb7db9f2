# round 5 A/B on one box: three-launch block (GBX_DATTN=1) vs five-launch block, driver-style 20 steps and 128 steps, 8B S22 / S40 / 3B S40
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 $2: "; env $1 python bench.py $2 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; }
for s in "--steps 20 --warmup 8" "--steps 128 --warmup 8"; do
  for v in GBX_DATTN=0 GBX_DATTN=1 GBX_DATTN=0 GBX_DATTN=1; do run "$v" "$s"; done
done
for v in GBX_DATTN=0 GBX_DATTN=1; do run "$v" "--steps 20 --warmup 8 --strategy S40"; done
for v in GBX_DATTN=0 GBX_DATTN=1; do run "$v" "--steps 20 --warmup 8 --strategy S40 --model llama3.2-3b"; done
