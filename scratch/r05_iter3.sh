cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_dattn_gpu.py tests/test_runner_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { echo -n "$1 $2: "; env $1 python bench.py $2 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us')"; }
for v in ${VARIANTS:-GBX_DATTN_MX=0 GBX_DATTN_MX=1 GBX_DATTN_MX=0 GBX_DATTN_MX=1}; do run "$v" "--steps 20 --warmup 8"; done
if [ -z "$NO_TIMELINE" ]; then
GBX_QLINEAR_LIB=$PWD/scratch/stamps/libgbx_stamps.so python3 scratch/layer_timeline2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/layer_timeline2.txt
sed -n 1,34p gpurun_out/r05/layer_timeline2.txt
fi
