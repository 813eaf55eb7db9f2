cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
LAYERS=0,1 CONTEXT=128 CAPACITY=768 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/tl_dmlpv4.txt
sed -n 1,3p gpurun_out/r06/tl_dmlpv4.txt; grep -A12 "^-- dmlp" gpurun_out/r06/tl_dmlpv4.txt; tail -17 gpurun_out/r06/tl_dmlpv4.txt
