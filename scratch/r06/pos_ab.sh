cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py tests/test_model_gpu.py tests/test_handoff_fault_gpu.py -x -q -m gpu 2>&1 | tail -3
one() { python bench.py --strategy $1 --context $2 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"; }
for cfg in "S40 128" "S30 128" "S22 128" "S22 600" "S40 600"; do
  set -- $cfg
  echo -n "$1 C=$2  new: "; one $1 $2; one $1 $2; echo -n "  old: "; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so one $1 $2; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so one $1 $2; echo
done
echo -n "3B S40 new: "; python bench.py --model llama3.2-3b --strategy S40 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"
echo -n " old: "; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so python bench.py --model llama3.2-3b --strategy S40 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"
