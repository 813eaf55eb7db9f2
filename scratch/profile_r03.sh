# Round-3 profile collection on the GPU box (one rocprofv3 run per pass; --pmc passes carry --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 128 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b32 -- python3 $R/bench.py --batch 32 --no-cpu-baseline --no-kernel-leg --no-extra --steps 16 --warmup 2 --context 128 > $O/stats_b32.log 2>&1
for w in qkv o; do
  WHICH=$w timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_$w -- python3 $R/scratch/one_short.py > $O/pmc_$w.log 2>&1
  WHICH=$w timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- python3 $R/scratch/one_short.py > $O/fetch_$w.log 2>&1
done
cd $R
python3 profiles/summarize.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 24 > $O/kernel_stats.txt 2>&1
python3 profiles/summarize.py $(ls $O/stats_b32/*/*_kernel_stats.csv | head -1) 24 > $O/b32_kernel_stats.txt 2>&1
tail -1 $O/stats.log | cut -c1-3000 > $O/bench_line_profiled.txt
python3 - <<'PY' > $O/pmc_gemv_short.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_r03"
print("# rocprofv3 --pmc (separate passes, --kernel-trace only; scratch/profile_r03.sh over scratch/one_short.py): the SHORT decode launches,")
print("# per-launch means; q|k|v = fused RMSNorm + [4096 | 1024 | 1024] x 4096 2-bit (7.1 MB), o = [4096 x 4096] + residual (4.7 MB)")
for w in ("qkv", "o"):
    for n in ("pmc", "fetch"):
        f=glob.glob(f"{O}/{n}_{w}/*/*_counter_collection.csv")
        if not f: print(w, n, "missing"); continue
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "gemv_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for cn,v in sorted(acc.items()): print(f"{w:4s} {cn:24s} dispatches {len(v):4d} mean {sum(v)/len(v):.5g}" + ("   (x2 on gfx950 for bytes)" if cn == "FETCH_SIZE" else ""))
        t=glob.glob(f"{O}/{n}_{w}/*/*_kernel_trace.csv")
        if t and n == "pmc":
            d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if "gemv_kernel" in r["Kernel_Name"]]
            print(f"{w:4s} duration_us mean {sum(d)/len(d):.2f} n {len(d)}")
PY
# chain heads: the short launches with null weights (same instruction stream, no weight stream), back to back in a graph
for w in qkv o gateup down; do
  for fl in 0 2; do
    if [ $fl = 0 ]; then unset GBX_CHAIN_FLOOR; else export GBX_CHAIN_FLOOR=$fl; fi
    echo "chain_floor_mode=$fl $(WHICH=$w python3 scratch/one_short.py 2>/dev/null | tail -1)"
  done
done > $O/launch_heads.txt 2>&1
unset GBX_CHAIN_FLOOR
python3 bench.py > $O/bench.json 2> $O/bench.err
rm -rf $O/stats $O/stats_b32 $O/pmc_qkv $O/pmc_o $O/fetch_qkv $O/fetch_o
ls -la $O
