# Round-4 PMC passes (one rocprofv3 run per pass; --pmc passes carry --kernel-trace only): the four decode GEMV launches of a 2-bit
# 8B layer, previous kernel (gbx_qlinear_fused_forward) vs decode-block GEMV (gbx_dgemv_forward), instruction / wait counters and
# HBM traffic; plus the null-weight chain floor of the new path and the launch heads.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
for w in gateup qkv o down; do for impl in old new; do
  WHICH=$w IMPL=$impl CARRIED=1 timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_${w}_$impl -- python3 $R/scratch/dg_launch.py > $O/pmc_${w}_$impl.log 2>&1
done; done
WHICH=gateup IMPL=new CARRIED=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_gateup -- python3 $R/scratch/dg_launch.py > $O/fetch_gateup.log 2>&1
WHICH=gateup IMPL=new CARRIED=1 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_gateup -- python3 $R/scratch/dg_launch.py > $O/write_gateup.log 2>&1
cd $R
python3 - <<'PY' > $O/pmc_gemv.txt 2>&1
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_r04"
print("# rocprofv3 --pmc (separate passes, --kernel-trace only; scratch/profile_r04b.sh over scratch/dg_launch.py): the four decode GEMV launches of a 2-bit")
print("# Llama-3-8B layer, per-launch means over graph replays that cycle the weights through > 256 MiB.  old = gbx::gemv_kernel (gbx_qlinear_fused_forward,")
print("# round 3), new = gbx::dgemv_kernel (gbx_dgemv_forward, round 4, statistic carried).  SQ_* are summed over all waves of a launch.")
for w in ("gateup", "qkv", "o", "down"):
    for impl in ("old", "new"):
        f=glob.glob(f"{O}/pmc_{w}_{impl}/*/*_counter_collection.csv")
        if not f: print(w, impl, "missing"); continue
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "gemv_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        t=glob.glob(f"{O}/pmc_{w}_{impl}/*/*_kernel_trace.csv")
        d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if "gemv_kernel" in r["Kernel_Name"]] if t else [0]
        print(f"{w:7s} {impl:3s} duration_us {sum(d)/max(len(d),1):6.2f}  " + "  ".join(f"{cn} {sum(v)/len(v):.4g}" for cn,v in sorted(acc.items())))
for n in ("fetch", "write"):
    f=glob.glob(f"{O}/{n}_gateup/*/*_counter_collection.csv")
    if not f: print(n, "missing"); continue
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "dgemv_kernel" in r["Kernel_Name"]]
    print(f"gateup new {n.upper()}_SIZE mean {sum(v)/len(v):.6g} over {len(v)} dispatches" + ("   (x2 on gfx950 for bytes of wide streaming reads; the counter unit is per the rocprofv3 definition: KB)" if n == "fetch" else ""))
PY
cat $O/pmc_gemv.txt
for w in qkv o gateup down; do for fl in 0 2; do
  if [ $fl = 0 ]; then unset GBX_CHAIN_FLOOR; else export GBX_CHAIN_FLOOR=$fl; fi
  echo "chain_floor_mode=$fl $(WHICH=$w IMPL=new CARRIED=1 python3 scratch/dg_launch.py 2>/dev/null | tail -1)"
done; done > $O/launch_heads.txt 2>&1
unset GBX_CHAIN_FLOOR
cat $O/launch_heads.txt
for fl in 0 2; do if [ $fl = 0 ]; then unset GBX_CHAIN_FLOOR; else export GBX_CHAIN_FLOOR=$fl; fi
  echo "chain_floor_mode=$fl $(python3 bench.py --steps 128 --no-extra --no-cpu-baseline --no-kernel-leg 2>/dev/null | tail -1 | cut -c1-160)"; done > $O/chain_floor.txt
unset GBX_CHAIN_FLOOR
cat $O/chain_floor.txt
rm -rf $O/pmc_*_old $O/pmc_*_new $O/fetch_gateup $O/write_gateup
