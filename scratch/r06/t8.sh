cd $GRAFT_REPO_ROOT
CONTEXT=128 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline_block.py 2>&1 | grep -v amdgpu.ids | head -20
