cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_dblock_gpu.py tests/test_runner_fullsize_gpu.py tests/test_model_gpu.py tests/test_handoff_fault_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
echo -n "512 steps dblock: "; python bench.py --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"
echo -n "512 steps two-launch: "; GBX_DBLOCK=0 python bench.py --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"
done
