cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
GBX_QLINEAR_LIB=$PWD/scratch/stamps/libgbx_stamps.so python3 scratch/layer_timeline3.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/layer_timeline3.txt
tail -40 gpurun_out/r05/layer_timeline3.txt
