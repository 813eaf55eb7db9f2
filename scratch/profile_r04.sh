# Round-4 profile collection on the GPU box (one rocprofv3 run per pass; --pmc passes carry --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 128 > $O/stats.log 2>&1
cd $R
python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 24 > $O/kernel_stats.txt 2>&1
tail -1 $O/stats.log | cut -c1-3000 > $O/bench_line_profiled.txt
rm -rf $O/stats
cat $O/kernel_stats.txt
