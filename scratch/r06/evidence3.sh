# final evidence of round 6 on the final code: three full-suite passes, smoke, kernel stats at the driver's flags, the driver-style bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
: > $O/full_suite_passes.txt
for i in 1 2 3; do
  echo "=== full -m gpu suite, pass $i ($(date -u +%H:%M:%S))" >> $O/full_suite_passes.txt
  timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:randomly 2>&1 | tail -4 >> $O/full_suite_passes.txt
done
cat $O/full_suite_passes.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fstats1 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-kernel-leg > $O/fstats1.log 2>&1
python3 $R/profiles/summarize.py $(ls $O/fstats1/*/*_kernel_stats.csv | head -1) 14 > $O/kernel_stats_steps20.txt 2>&1
rm -rf $O/fstats1
cut -c1-175 $O/kernel_stats_steps20.txt
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python -c "
import json; d=json.loads(open('$O/bench_driver_style.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_launch']); print({k:(v.get('tok_s') if isinstance(v,dict) and 'tok_s' in v else (v.get('decode',{}).get('tok_s') if isinstance(v,dict) else None)) for k,v in d['extra'].items()}); print(d['cpu_baseline']['value'])"
python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default 512 steps:', d['value'], d['extra']['generate_step_S22']['tok_s'])"
