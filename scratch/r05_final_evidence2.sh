# final code, again: the driver-style line and the kernel stats of the same default command
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python bench.py > $O/bench_driver_style.json 2> $O/bench_driver_style.err
tail -c 300 $O/bench_driver_style.json
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fstats1 -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-kernel-leg > $O/fstats1.log 2>&1
python3 $R/profiles/summarize.py $(ls $O/fstats1/*/*_kernel_stats.csv | head -1) 14 > $O/kernel_stats_final_dattn1.txt 2>&1
rm -rf $O/fstats1
cut -c1-175 $O/kernel_stats_final_dattn1.txt
