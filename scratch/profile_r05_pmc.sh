# HBM traffic of the dominant launches of the two-launch block (dmlp_kernel, dattn_kernel): FETCH_SIZE and WRITE_SIZE in separate passes,
# --kernel-trace only, over the decode bench itself (20 steps; the token's 27 two-bit layers cycle 0.9 GB of weights)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/scratch/dm_tokens.py 4 > $O/pmc_$c.log 2>&1
done
cd $R
python3 - <<'PY' > $O/pmc_dmlp.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r05"
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only; scratch/profile_r05_pmc.sh over scratch/dm_tokens.py: 4 eagerly launched tokens of 8B S22, default switches)")
print("# per-launch means over the token graph's replays (counter unit KB)")
for n in ("FETCH_SIZE", "WRITE_SIZE"):
    f=glob.glob(f"{O}/pmc_{n}/*/*_counter_collection.csv")
    if not f: print(n, "missing"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "dmlp_kernel" in k or "dattn_kernel<2" in k or "dense_gemv" in k: acc[k.split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f"{n} {k}: mean {sum(v)/len(v):.4f} over {len(v)} dispatches")
print("# bytes per launch = 2 x FETCH_SIZE KB (gfx950 correction for wide streaming reads) + WRITE_SIZE KB")
PY
cat $O/pmc_dmlp.txt; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
