# round 5: kernel stats of the default decode bench, two-launch blocks (GBX_DMLP=1) vs three-launch (same box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
for v in 1 0; do
  GBX_DMLP=$v timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsm$v -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-kernel-leg --steps 20 --warmup 8 > $O/statsm$v.log 2>&1
  python3 $R/profiles/summarize.py $(ls $O/statsm$v/*/*_kernel_stats.csv | head -1) 12 > $O/kernel_stats_dmlp$v.txt 2>&1
  rm -rf $O/statsm$v
  cut -c1-175 $O/kernel_stats_dmlp$v.txt
done
