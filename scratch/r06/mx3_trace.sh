cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
for m in 2 0; do
  export GBX_DMLP_MX3=$m
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$m -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-kernel-leg > $O/tr_$m.log 2>&1
  python3 - <<PY
import csv, glob, collections
f=glob.glob("$O/tr_$m/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"].split("(")[0].replace("void gbx::","")[:28] for r in rows]
# a token = from one embed_next_kernel to the next; take the last 12 tokens
idx=[i for i,n in enumerate(names) if n.startswith("embed_next")]
toks=[(idx[i],idx[i+1]) for i in range(len(idx)-14,len(idx)-1)]
L=toks[0][1]-toks[0][0]
acc=collections.defaultdict(list); gaps=collections.defaultdict(list)
for a,b in toks:
    if b-a!=L: continue
    for k in range(a,b):
        acc[k-a].append((int(rows[k]["End_Timestamp"])-int(rows[k]["Start_Timestamp"]))/1e3)
        gaps[k-a].append((int(rows[k+1]["Start_Timestamp"])-int(rows[k]["End_Timestamp"]))/1e3)
print("== GBX_DMLP_MX3=$m  launches per token", L, " token span us", sum(sum(v)/len(v) for v in acc.values())+sum(sum(v)/len(v) for v in gaps.values()))
a,b=toks[0]
print(" ".join(f"{k}:{names[a+k][:10]}:{sum(acc[k])/len(acc[k]):.1f}+{sum(gaps[k])/len(gaps[k]):.1f}" for k in range(min(L,14))))
print(" ... ".join(f"{k}:{names[a+k][:10]}:{sum(acc[k])/len(acc[k]):.1f}+{sum(gaps[k])/len(gaps[k]):.1f}" for k in range(max(0,L-9),L)))
db=[sum(acc[k])/len(acc[k]) for k in range(L) if names[a+k].startswith("dblock")]
print("dblock launches: mean %.2f  first five %s  last three %s" % (sum(db)/len(db), [round(x,2) for x in db[:5]], [round(x,2) for x in db[-3:]]))
PY
  rm -rf $O/tr_$m
done
