"""List registers / spills / scratch of every kernel in a built library (the .hip_fatbin bundles of each translation unit).
usage: python scratch/kernel_resources.py gbx-lm_amd/csrc/libgbx_qlinear.so [--all]   (default: only kernels with spills or scratch)"""
import re, struct, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1]
show_all = "--all" in sys.argv
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "dummy")])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    rows = []
    for bi, st in enumerate(starts):
        n = struct.unpack_from("<Q", blob, st + 24)[0]
        off = st + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24: off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" not in triple or sz == 0:
                continue
            co = os.path.join(td, f"co{bi}.o")
            open(co, "wb").write(blob[st + o: st + o + sz])
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for km in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
                def f(k):
                    m = re.search(rf"\.{k}:\s*(\S+)", km)
                    return m.group(1) if m else "?"
                rows.append((f("name"), f("vgpr_count"), f("sgpr_count"), f("vgpr_spill_count"), f("sgpr_spill_count"), f("private_segment_fixed_size"), f("group_segment_fixed_size")))
    bad = 0
    for name, v, s, vs, ss, priv, lds in sorted(set(rows)):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110]
        flag = (vs not in ("0", "?")) or (priv not in ("0", "?"))
        bad += flag
        if show_all or flag:
            print(f"{dem:110s} vgpr {v:>3} sgpr {s:>3} vgpr_spill {vs:>3} sgpr_spill {ss:>3} scratch {priv:>4} B")
    print(f"{len(set(rows))} kernels, {bad} with vector spills or scratch")
