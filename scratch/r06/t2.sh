cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_dblock_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06/t2_dblock.txt
cat gpurun_out/r06/t2_dblock.txt
timeout 1500 python -m pytest tests/test_handoff_fault_gpu.py tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py tests/test_abi.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06/t1.txt
cat gpurun_out/r06/t1.txt
for v in 1 0; do echo "GBX_DBLOCK=$v"; GBX_DBLOCK=$v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | cut -c1-400; done
