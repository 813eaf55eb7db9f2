cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
for dg in 1 0; do
  GBX_DGEMV=$dg timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_s40_$dg -- python3 $R/bench.py --strategy S40 --no-cpu-baseline --no-extra --no-kernel-leg --steps 64 > $O/stats_s40_$dg.log 2>&1
  cd $R; echo "=== S40 GBX_DGEMV=$dg"; python3 profiles/summarize.py $(ls $O/stats_s40_$dg/*/*_kernel_stats.csv | head -1) 7 | cut -c1-150; rm -rf $O/stats_s40_$dg; cd /tmp
done
