cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:randomly 2>&1 | tail -5
one() { python bench.py --model $1 --strategy $2 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"; }
for cfg in "llama3-8b S22" "llama3-8b S40" "llama3-8b S30" "llama3.2-3b S40"; do set -- $cfg; echo -n "$1 $2: "; one $1 $2; one $1 $2; echo; done
