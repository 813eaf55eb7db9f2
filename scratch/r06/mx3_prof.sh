cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
for m in 1 0; do
  export GBX_DMLP_MX3=$m
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mx3_$m -- python3 $R/bench.py --strategy S22 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-kernel-leg > $O/mx3_$m.log 2>&1
  echo "== GBX_DMLP_MX3=$m"; grep -o '"value": [0-9.]*' $O/mx3_$m.log | head -1
  python3 $R/profiles/summarize.py $(ls $O/mx3_$m/*/*_kernel_stats.csv | head -1) 12 2>&1 | cut -c1-150
  rm -rf $O/mx3_$m
done
