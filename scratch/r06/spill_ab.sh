cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dattn_gpu.py tests/test_dgemv_gpu.py tests/test_runner_fullsize_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
one() { python bench.py --strategy $1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"; }
for S in S30 S22 S40; do
  echo -n "$S  new: "; one $S; one $S; echo -n "  old: "; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so one $S; GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so one $S; echo
done
echo -n "S30 context 150 new/old: "; python bench.py --strategy S30 --context 150 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' ')"
GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_old.so python bench.py --strategy S30 --context 150 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"
