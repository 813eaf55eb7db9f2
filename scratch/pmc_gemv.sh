cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemv
mkdir -p $O
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/a -- python3 $R/scratch/one_fused.py > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -- python3 $R/scratch/one_fused.py > $O/b.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_gemv"
for n in "ab":
    f=glob.glob(f"{O}/{n}/*/*_counter_collection.csv")
    if not f: print(n,"missing"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemv_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for cn,v in sorted(acc.items()): print(f"{cn:28s} dispatches {len(v)} mean {sum(v)/len(v):.5g}")
    t=glob.glob(f"{O}/{n}/*/*_kernel_trace.csv")
    if t:
        d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if "gemv_kernel" in r["Kernel_Name"]]
        print("duration_us mean", round(sum(d)/len(d),2), "n", len(d))
PY
rm -rf $O/a $O/b
