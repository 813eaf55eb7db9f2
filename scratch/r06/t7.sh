cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_dblock_gpu.py -x -q -m gpu > gpurun_out/r06/t7.txt 2>&1
echo rc=$? >> gpurun_out/r06/t7.txt
head -60 gpurun_out/r06/t7.txt | cut -c1-200
