# HBM traffic of the gate|up launch as the default dispatch runs it (dgemv_mx_kernel): FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
WHICH=gateup IMPL=new CARRIED=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_gateup_mx -- python3 $R/scratch/dg_launch.py > $O/fetch_gateup_mx.log 2>&1
WHICH=gateup IMPL=new CARRIED=1 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_gateup_mx -- python3 $R/scratch/dg_launch.py > $O/write_gateup_mx.log 2>&1
cd $R
python3 - <<'PY' > $O/pmc_gateup_mx.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_r04"
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only; scratch/profile_r04_pmc_mx.sh over scratch/dg_launch.py, default switches):")
print("# the gate|up launch of a 2-bit Llama-3-8B layer as gbx_dgemv_forward dispatches it, per-launch means over graph replays cycling > 256 MiB of weights")
for n in ("fetch", "write"):
    f=glob.glob(f"{O}/{n}_gateup_mx/*/*_counter_collection.csv")
    if not f: print(n, "missing"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "dgemv" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f"{n.upper()}_SIZE {k}: mean {sum(v)/len(v):.4f} over {len(v)} dispatches")
print("# bytes per launch = 2 x FETCH_SIZE KB (gfx950 correction for wide streaming reads) + WRITE_SIZE KB")
PY
cat $O/pmc_gateup_mx.txt; rm -rf $O/fetch_gateup_mx $O/write_gateup_mx
