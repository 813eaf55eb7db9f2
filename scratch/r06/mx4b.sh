cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_dattn_gpu.py -x -q -m gpu -k "dmlp" 2>&1 | tail -3
LAYERS=0 CONTEXT=128 CAPACITY=768 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so timeout 300 python scratch/layer_timeline2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/tl_mx4b.txt; sed -n 1,3p gpurun_out/r06/tl_mx4b.txt; grep -A12 "^-- dmlp" gpurun_out/r06/tl_mx4b.txt
one() { python bench.py --strategy $1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), end=' | ')"; }
for S in S40; do
  echo -n "$S  MX4=1: "; GBX_DMLP_MX4=1 one $S; GBX_DMLP_MX4=1 one $S; echo; echo -n "$S  MX4=0: "; GBX_DMLP_MX4=0 one $S; GBX_DMLP_MX4=0 one $S; echo
done
