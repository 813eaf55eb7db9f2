cd $GRAFT_REPO_ROOT
GBX_QLINEAR_LIB=$PWD/scratch/exp/lib_$1.so timeout 600 python -m pytest tests/test_dblock_gpu.py -x -q -m gpu 2>&1 | tail -2
bash scratch/r06/ab.sh "$@"
