cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06; CONTEXT=128 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline_block.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/tl_strag.txt
sed -n 36,200p gpurun_out/r06/tl_strag.txt
