cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_prefill
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/scratch/prefill_model.py 3 > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p -- python3 $R/scratch/prefill_model.py 1 > $O/p.log 2>&1
cd $R
python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 30 > $O/kernel_stats.txt 2>&1
python3 - <<'PY' > $O/pmc.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_prefill"
f=glob.glob(f"{O}/p/*/*_counter_collection.csv")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    kk = next((n for n in ("attn_prefill_kernel","gemm_dense_kernel","dequant_units_kernel") if n in k), None)
    if kk: acc[kk][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kk,c in acc.items():
    for cn,v in sorted(c.items()):
        print(f"{kk:22s} {cn:28s} dispatches {len(v):3d} mean {sum(v)/len(v):.4g}")
PY
tail -3 $O/stats.log > $O/passes.txt
rm -rf $O/stats $O/p
cat $O/kernel_stats.txt $O/pmc.txt $O/passes.txt
