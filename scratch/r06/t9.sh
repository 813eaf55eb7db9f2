cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_dattn_gpu.py -x -q -m gpu -k "dmlp" 2>&1 | tail -25 > gpurun_out/r06/t9.txt
tail -12 gpurun_out/r06/t9.txt
timeout 1200 python -m pytest tests/test_runner_fullsize_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -5
for s in S40; do for v in 1 0; do echo -n "8B $s GBX_DMLP=$v: "; GBX_DMLP=$v python bench.py --strategy $s --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"; done; done
for v in 1 0; do echo -n "3B S40 GBX_DMLP=$v: "; GBX_DMLP=$v python bench.py --model llama3.2-3b --strategy S40 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"; done
for v in 1 0; do echo -n "8B S22 GBX_DMLP=$v: "; GBX_DMLP=$v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s')"; done
