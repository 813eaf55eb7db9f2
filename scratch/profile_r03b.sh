# Round-3 additions: HBM traffic of the dominant launch (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) and the short-prompt TTFT table.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03b
mkdir -p $O
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/scratch/one_fused.py > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/scratch/one_fused.py > $O/write.log 2>&1
cd $R
python3 - <<'PY' > $O/pmc_traffic.txt 2>&1
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_r03b"
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over scratch/one_fused.py: the fused RMSNorm + 2-bit gate|up + SwiGLU")
print("# GEMV launch (2 x [14336, 4096], g = 128, packed scale|zero companion), round-3 kernel.  FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB on")
print("# this stack; bytes = value * 1024, FETCH x2 on gfx950 (MI355X_MICROARCH, HBM section).  Algorithmic bytes per launch: 33,075,200.")
tot = 0
for name in ("fetch","write"):
    f=glob.glob(f"{O}/{name}/*/*_counter_collection.csv")
    if not f: print(name,"missing"); continue
    vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gemv_kernel" in r["Kernel_Name"]]
    print(name, "gemv_kernel dispatches", len(vals), "mean", sum(vals)/max(1,len(vals)), "min", min(vals), "max", max(vals))
PY
timeout 900 python3 scratch/ttft.py > $O/ttft.txt 2>&1
rm -rf $O/fetch $O/write
