cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for cp in 1 0; do
echo "=== GBX_DBLOCK_CODE_PREFETCH=$cp"
GBX_DBLOCK_CODE_PREFETCH=$cp CONTEXT=128 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline_block.py 2>&1 | grep -v amdgpu.ids | head -15
done > gpurun_out/r06/tl_block_cp.txt
cat gpurun_out/r06/tl_block_cp.txt
for v in 1 0 1 0; do echo "GBX_DBLOCK_CODE_PREFETCH=$v"; GBX_DBLOCK_CODE_PREFETCH=$v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | cut -c1-120; done
