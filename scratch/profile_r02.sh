# Round-2 profile collection on the GPU box (one rocprofv3 run per pass; --pmc passes carry --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extra > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b32 -- python3 $R/bench.py --batch 32 --no-cpu-baseline --no-kernel-leg --no-extra --steps 16 --warmup 2 --context 128 > $O/stats_b32.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/scratch/one_fused.py > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/scratch/one_fused.py > $O/write.log 2>&1
cd $R
python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 24 > $O/kernel_stats.txt 2>&1
python3 profiles/summarize.py $(ls $O/stats_b32/*/*_kernel_stats.csv | head -1) 24 > $O/b32_kernel_stats.txt 2>&1
tail -1 $O/stats.log | cut -c1-3000 > $O/bench_line.txt
python3 - <<'PY' > $O/pmc.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_r02"
for name in ("fetch","write"):
    f=glob.glob(f"{O}/{name}/*/*_counter_collection.csv")
    if not f: print(name,"missing"); continue
    vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gemv_kernel" in r["Kernel_Name"]]
    print(name, "gemv_kernel dispatches", len(vals), "mean", sum(vals)/max(1,len(vals)), "min", min(vals), "max", max(vals))
for name in ():
    f=glob.glob(f"{O}/{name}/*/*_counter_collection.csv")
    if not f: print(name,"missing"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        kk = "gemm_dense_kernel" if "gemm_dense_kernel" in k else "dequant_kernel" if "dequant_kernel" in k else "qgemm_prefill_kernel" if "qgemm_prefill" in k else None
        if kk: acc[kk][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kk,c in acc.items():
        for cn,v in sorted(c.items()):
            print(f"{name} {kk:22s} {cn:28s} dispatches {len(v):3d} mean {sum(v)/len(v):.4g}")
    # kernel durations of the same run
    t=glob.glob(f"{O}/{name}/*/*_kernel_trace.csv")
    if t:
        d=collections.defaultdict(list)
        for r in csv.DictReader(open(t[0])):
            k=r["Kernel_Name"]
            kk = "gemm_dense_kernel" if "gemm_dense_kernel" in k else "dequant_kernel" if "dequant_kernel" in k else None
            if kk: d[kk].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
        for kk,v in d.items(): print(f"{name} {kk:22s} duration_us mean {sum(v)/len(v):.1f} n {len(v)}")
PY
rm -rf $O/stats $O/stats_b32 $O/fetch $O/write
ls -la $O
