# round 5: kernel stats of the default decode bench with the three-launch block and with the five-launch block (same box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
for v in 1 0; do
  GBX_DATTN=$v timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$v -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-kernel-leg --steps 64 > $O/stats$v.log 2>&1
  python3 $R/profiles/summarize.py $(ls $O/stats$v/*/*_kernel_stats.csv | head -1) 14 > $O/kernel_stats_dattn$v.txt 2>&1
  rm -rf $O/stats$v
  cut -c1-175 $O/kernel_stats_dattn$v.txt
done
