cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py tests/test_stream_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { echo -n "$1 $2: "; env $1 python bench.py $2 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; }
run GBX_DATTN=1 "--steps 20 --warmup 8"
run GBX_DATTN=1 "--warmup 8"
run GBX_DATTN=1 "--steps 20 --warmup 8"
run GBX_DATTN=1 "--warmup 8"
CONTEXT=128 CAPACITY=768 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so python scratch/layer_timeline2.py > gpurun_out/r05/tl2_128.txt 2>&1
