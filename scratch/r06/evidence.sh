# round-6 evidence: three full-suite passes (no retry wrapper), kernel stats at the driver's exact flags, HBM traffic of the one-launch block
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
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fstats1 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-kernel-leg > $O/fstats1.log 2>&1
python3 $R/profiles/summarize.py $(ls $O/fstats1/*/*_kernel_stats.csv | head -1) 14 > $O/kernel_stats_steps20.txt 2>&1
rm -rf $O/fstats1
cut -c1-175 $O/kernel_stats_steps20.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/scratch/dm_tokens.py 4 > $O/pmc_$c.log 2>&1
done
cd $R
python3 - <<'PY' > $O/pmc_dblock.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r06"
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only; scratch/r06/evidence.sh over scratch/dm_tokens.py: 4 eagerly launched tokens of 8B S22, default switches)")
print("# per-launch means (counter unit KB)")
for n in ("FETCH_SIZE", "WRITE_SIZE"):
    f=glob.glob(f"{O}/pmc_{n}/*/*_counter_collection.csv")
    if not f: print(n, "missing"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "dblock_kernel" in k or "dmlp_kernel" in k or "dattn_kernel<2" in k or "dense_gemv" in k: acc[k.split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f"{n} {k}: mean {sum(v)/len(v):.4f} over {len(v)} dispatches")
print("# bytes per launch = 2 x FETCH_SIZE KB (gfx950 correction for wide streaming reads) + WRITE_SIZE KB")
PY
cat $O/pmc_dblock.txt; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
tail -c 1500 $O/bench_driver_style.json
