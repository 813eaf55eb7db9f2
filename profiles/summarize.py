"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table (kernel names shortened)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# torch's own kernels (synthetic weight generation, elementwise setup: everything under at::native) are not part of the decode path
torch_rows = [r for r in rows if "at::native" in r["Name"] or "at::cuda" in r["Name"]]
rows = [r for r in rows if r not in torch_rows]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"<.*", "", n) if len(n) > 90 and not n.startswith("void gbx::") else n
    return n[:110]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {len(torch_rows)} torch (at::native) kernels left out: {sum(float(r['TotalDurationNs']) for r in torch_rows) / 1e3:.0f} us of model set-up")
print(f"{'kernel':110s} {'calls':>8s} {'total_us':>12s} {'avg_us':>9s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{short(r['Name']):110s} {r['Calls']:>8s} {float(r['TotalDurationNs'])/1e3:12.1f} {float(r['AverageNs'])/1e3:9.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
