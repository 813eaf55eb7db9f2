cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s40
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --strategy S40 --no-cpu-baseline --no-extra --steps 128 > $O/stats.log 2>&1
cd $R
python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 12 > $O/kernel_stats.txt 2>&1
for fl in 0 2; do
  if [ $fl = 0 ]; then unset GBX_CHAIN_FLOOR; else export GBX_CHAIN_FLOOR=$fl; fi
  python3 bench.py --strategy S40 --no-cpu-baseline --no-extra --steps 256 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain_floor_mode=$fl', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline'])"
done > $O/floor.txt 2>&1
rm -rf $O/stats
cat $O/kernel_stats.txt $O/floor.txt
