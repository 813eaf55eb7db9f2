# round 5, final code: (1) the driver-style bench line, (2) rocprofv3 kernel stats of the same default command (two-launch blocks) and of the
# round-4 five-launch structure on the same box, (3) the in-kernel stamp timelines (ctx 128, and ctx 500 on the cooperative sweep),
# (4) decode throughput against the cached context, fused launches vs five launches.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python bench.py > $O/bench_driver_style.json 2> $O/bench_driver_style.err
tail -c 600 $O/bench_driver_style.json
cd /tmp
for v in 1 0; do
  GBX_DATTN=$v timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fstats$v -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-kernel-leg > $O/fstats$v.log 2>&1
  python3 $R/profiles/summarize.py $(ls $O/fstats$v/*/*_kernel_stats.csv | head -1) 14 > $O/kernel_stats_final_dattn$v.txt 2>&1
  rm -rf $O/fstats$v
  cut -c1-175 $O/kernel_stats_final_dattn$v.txt
done
cd $R
CONTEXT=128 CAPACITY=768 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so python scratch/layer_timeline2.py > $O/layer_timeline2_ctx128.txt 2>&1
CONTEXT=500 CAPACITY=768 GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so python scratch/layer_timeline2.py > $O/layer_timeline2_ctx500_coop.txt 2>&1
run() { echo -n "$1 $2: "; env $1 python bench.py $2 --no-extra --no-cpu-baseline --no-kernel-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000,1), 'us per token')"; }
{
echo "# decode tok/s against the cached context (8B S22 unless named; --steps 32 --warmup 8; one box, back to back)"
for c in 28 128 256 384 512 1024 2048 4096 8192 16384; do
  run GBX_DATTN=1 "--context $c --steps 32 --warmup 8"
  run GBX_DATTN=0 "--context $c --steps 32 --warmup 8"
done
run GBX_DATTN=1 "--context 4096 --steps 32 --warmup 8 --model llama3.2-3b --strategy S40"
run GBX_DATTN=0 "--context 4096 --steps 32 --warmup 8 --model llama3.2-3b --strategy S40"
run GBX_DATTN=1 "--context 4096 --steps 32 --warmup 8 --strategy S40"
run GBX_DATTN=0 "--context 4096 --steps 32 --warmup 8 --strategy S40"
} > $O/context_sweep.txt 2>&1
cat $O/context_sweep.txt
