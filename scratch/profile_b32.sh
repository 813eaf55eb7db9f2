cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_b32; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --batch 32 --no-cpu-baseline --no-kernel-leg --no-extra --steps 16 --warmup 2 --context 128 > $O/log.txt 2>&1
cd $R; python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 30 > $O/kernel_stats.txt 2>&1; rm -rf $O/stats
