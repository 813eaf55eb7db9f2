cd $GRAFT_REPO_ROOT
export GBX_QLINEAR_LIB=$PWD/scratch/stamps/libgbx_stamps.so
for w in qkv o gateup down; do
  python3 scratch/stamps_dg.py $w old 2>/dev/null | grep -v amdgpu.ids
  python3 scratch/stamps_dg.py $w new 2>/dev/null | grep -v amdgpu.ids
  GBX_DG_R=2 python3 scratch/stamps_dg.py $w new 2>/dev/null | grep -v amdgpu.ids
  CARRIED=1 python3 scratch/stamps_dg.py $w new 2>/dev/null | grep -v amdgpu.ids
done
