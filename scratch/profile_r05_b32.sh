# round 5: kernel stats of the B = 32 decode step (cfg5@TP1), chip-wide slice-sum + norm launch on / off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
for v in 1 0; do
  GBX_NORM_XCD=$v timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b32stats$v -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-kernel-leg --batch 32 --steps 32 > $O/b32stats$v.log 2>&1
  python3 $R/profiles/summarize.py $(ls $O/b32stats$v/*/*_kernel_stats.csv | head -1) 12 > $O/b32_kernel_stats_xcd$v.txt 2>&1
  rm -rf $O/b32stats$v
  cut -c1-175 $O/b32_kernel_stats_xcd$v.txt
done
