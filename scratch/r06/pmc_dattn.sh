# verdict item 5(c): FETCH_SIZE / WRITE_SIZE of the fused attention launches (two-launch block, GBX_DBLOCK=0) at ~128 cached positions
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  GBX_DBLOCK=0 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcA_$c -- python3 $R/scratch/dm_tokens.py 136 > $O/pmcA_$c.log 2>&1
done
cd $R
python3 - <<'PY' > $O/pmc_dattn.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r06"
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only; scratch/r06/pmc_dattn.sh over scratch/dm_tokens.py 136 with GBX_DBLOCK=0:")
print("# 136 eagerly launched tokens of 8B S22 from an empty cache; means over the LAST 8 tokens = 128-135 cached positions; counter unit KB)")
res={}
for n in ("FETCH_SIZE", "WRITE_SIZE"):
    f=glob.glob(f"{O}/pmcA_{n}/*/*_counter_collection.csv")
    if not f: print(n, "missing"); continue
    rows=[r for r in csv.DictReader(open(f[0])) if "dattn_kernel" in r["Kernel_Name"] or "dmlp_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Dispatch_Id"]))
    acc=collections.defaultdict(list)
    for r in rows: acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        per_tok=len(v)//136
        last=v[-8*per_tok:]
        first=v[:8*per_tok]
        res[(n,k)]=sum(last)/len(last)
        print(f"{n} {k}: last 8 tokens mean {sum(last)/len(last):.1f} over {len(last)} dispatches; first 8 tokens (0-7 cached positions) mean {sum(first)/len(first):.1f}")
print("# bytes per launch = 2 x FETCH_SIZE KB (gfx950 correction for wide streaming reads) + WRITE_SIZE KB")
ks=sorted({k for (_,k) in res})
for k in ks:
    if ("FETCH_SIZE",k) in res and ("WRITE_SIZE",k) in res:
        print(f"{k}: {(2*res[('FETCH_SIZE',k)]+res[('WRITE_SIZE',k)])*1024/1e6:.2f} MB per launch")
PY
cat $O/pmc_dattn.txt; rm -rf $O/pmcA_FETCH_SIZE $O/pmcA_WRITE_SIZE
