# Round-1 profile collection on the GPU box (one rocprofv3 run per pass, each under its own timeout).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/scratch/one_fused.py > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/scratch/one_fused.py > $O/write.log 2>&1
cd $R
python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 24 > $O/kernel_stats.txt 2>&1
tail -2 $O/stats.log | cut -c1-1500 > $O/bench_line.txt
python3 - <<'PY' > $O/pmc.txt 2>&1
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_final"
for name in ("fetch","write"):
    f=glob.glob(f"{O}/{name}/*/*_counter_collection.csv")
    if not f: print(name,"missing"); continue
    vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gemv_kernel" in r["Kernel_Name"]]
    names=set(r["Kernel_Name"][:80] for r in csv.DictReader(open(f[0])) if "gemv_kernel" in r["Kernel_Name"])
    print(name, "dispatches", len(vals), "mean", sum(vals)/max(1,len(vals)), "min", min(vals), "max", max(vals), names)
PY
rm -rf $O/stats $O/fetch $O/write
ls -la $O
