# Final round-4 refresh: kernel stats of the default decode bench and of the B = 32 step, and the bench lines themselves.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 128 > $O/stats.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b32 -- python3 $R/bench.py --batch 32 --no-cpu-baseline --no-kernel-leg --no-extra --steps 16 --warmup 2 --context 128 > $O/stats_b32.log 2>&1
cd $R
python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 24 > $O/kernel_stats.txt 2>&1
python3 profiles/summarize.py $(ls $O/stats_b32/*/*_kernel_stats.csv | head -1) 16 > $O/b32_kernel_stats.txt 2>&1
rm -rf $O/stats $O/stats_b32
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_style.json
python3 bench.py --batch 32 --steps 32 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-leg 2>/dev/null | tail -1 > $O/bench_b32.json
head -12 $O/kernel_stats.txt | cut -c1-170; head -10 $O/b32_kernel_stats.txt | cut -c1-170
python3 -c "
import json
for f in ('bench_driver_style','bench_b32'):
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['value'],1), d['unit'], round(d['ms_per_step'],4), d.get('roofline'))
"
