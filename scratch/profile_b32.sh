cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b32 -- python3 $R/bench.py --batch 32 --no-cpu-baseline --no-kernel-leg --no-extra --steps 16 --warmup 2 --context 128 > $O/stats_b32.log 2>&1
cd $R; python3 profiles/summarize.py $(ls $O/stats_b32/*/*_kernel_stats.csv | head -1) 16 > $O/b32_kernel_stats.txt; cat $O/b32_kernel_stats.txt | cut -c1-170; tail -1 $O/stats_b32.log | cut -c1-200; rm -rf $O/stats_b32
