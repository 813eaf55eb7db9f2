cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_dblock_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06/t3_dblock.txt
cat gpurun_out/r06/t3_dblock.txt | tail -8
CONTEXT=128 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline_block.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/tl_block.txt
cat gpurun_out/r06/tl_block.txt
for v in 1 0 1 0; do echo "GBX_DBLOCK=$v"; GBX_DBLOCK=$v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | cut -c1-120; done
