cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_dblock_gpu.py tests/test_handoff_fault_gpu.py -x -q -m gpu 2>&1 | tail -3
CONTEXT=128 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline_block.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/tl_block_final.txt
cat gpurun_out/r06/tl_block_final.txt | head -3
CONTEXT=128 CAPACITY=768 GBX_DBLOCK=0 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/tl_two_launch.txt
head -4 gpurun_out/r06/tl_two_launch.txt
for v in "GBX_DBLOCK=1" "GBX_DBLOCK=0" "GBX_DBLOCK=1" "GBX_DBLOCK=0" "GBX_DBLOCK=1" "GBX_DBLOCK=0"; do echo -n "$v: "; env $v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us/token')"; done | tee gpurun_out/r06/ab_final.txt
