cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
for mx in 0 1 2; do
  GBX_DG_MX=$mx timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mx$mx -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-kernel-leg --steps 128 > $O/stats_mx$mx.log 2>&1
  cd $R; echo "=== GBX_DG_MX=$mx"; python3 profiles/summarize.py $(ls $O/stats_mx$mx/*/*_kernel_stats.csv | head -1) 8 | cut -c1-150; tail -1 $O/stats_mx$mx.log | cut -c1-120; rm -rf $O/stats_mx$mx; cd /tmp
done
