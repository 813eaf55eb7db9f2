# Memory-side counters of the batch GEMM (M = 32, 28672 x 4096, 2 bits): HBM fetch / write bytes and the L2 hit rate -- separate passes, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_batch_mem
mkdir -p $O
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/a -- python3 $R/scratch/one_batch.py 28672 4096 > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/b -- python3 $R/scratch/one_batch.py 28672 4096 > $O/b.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/c -- python3 $R/scratch/one_batch.py 28672 4096 > $O/c.log 2>&1
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $O/d -- python3 $R/scratch/one_batch.py 28672 4096 > $O/d.log 2>&1
cd $R
python3 - <<'PY' > $O/pmc_batch_mem.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_batch_mem"
print("# rocprofv3 --pmc (separate passes, --kernel-trace only; scratch/pmc_batch_mem.sh over scratch/one_batch.py): qgemm_batch_kernel at M = 32, N = 28672, K = 4096, 2 bits,")
print("# 5 launches on the same weights (the first is cold).  Algorithmic bytes per launch: packed words 29.36 MB + scales / zeros 3.67 MB + x 0.26 MB + y 1.84 MB.")
for n in "abcd":
    f=glob.glob(f"{O}/{n}/*/*_counter_collection.csv")
    if not f: print(n,"missing (counter not available?)"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "qgemm_batch" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for cn,v in sorted(acc.items()): print(f"{cn:30s} per launch: " + " ".join(f"{x:.6g}" for x in v))
PY
cat $O/pmc_batch_mem.txt; tail -3 $O/c.log | cut -c1-200; rm -rf $O/a $O/b $O/c $O/d
