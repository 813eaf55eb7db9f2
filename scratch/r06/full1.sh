cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:randomly 2>&1 | tail -12
python scratch/gen_rate.py S22 300 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | cut -c1-200
