cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:randomly --deselect tests/test_model_gpu.py 2>&1 | tail -5
