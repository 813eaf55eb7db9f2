cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_generate_fast_gpu.py tests/test_handoff_fault_gpu.py tests/test_model_gpu.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -15
python scratch/gen_rate2.py 2>&1 | grep -v amdgpu.ids | head -7
