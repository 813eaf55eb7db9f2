cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dgemv_gpu.py -q 2>&1 | tail -4
bash scratch/dg_sweep.sh 2>&1 | tee gpurun_out/dg_sweep4.txt
python bench.py --steps 64 --warmup 8 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | cut -c1-200
GBX_DG_SSQ=0 python bench.py --steps 64 --warmup 8 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | cut -c1-200
GBX_DGEMV=0 python bench.py --steps 64 --warmup 8 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | cut -c1-200
GBX_QLINEAR_LIB=$PWD/scratch/stamps/libgbx_stamps.so python scratch/layer_timeline.py 2>&1 | grep -v amdgpu.ids | head -40 | tee gpurun_out/timeline4.txt
