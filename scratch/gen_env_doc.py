"""Regenerates the table of DESIGN.md section 8 from gbx-lm_amd/env.py:TABLE and csrc/gbx_env.h:env_table (tests/test_env_table.py checks it)."""
import importlib, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
env = importlib.import_module("gbx_lm_amd.env")
src = open(os.path.join(ROOT, "gbx-lm_amd", "csrc", "gbx_env.h")).read()
rows = ["| switch | default | read by | effect |", "|---|---|---|---|"]
for name, (dflt, doc) in env.TABLE.items():
    rows.append(f"| `{name}` | {dflt if dflt != '' else '—'} | host (`gbx-lm_amd/env.py`) | {doc.replace('|', chr(92) + '|')} |")
for m in re.finditer(r'\{"(GBX_[A-Z0-9_]+)",\s*(-?\d+),\s*"((?:[^"\\]|\\.)*)"\}', src):
    doc = m.group(3).replace('\\"', '"').replace("|", "\\|")
    rows.append(f"| `{m.group(1)}` | {m.group(2)} | kernels (`csrc/gbx_env.h`) | {doc} |")
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("| switch | default | read by | effect |")
b = a
lines = s[a:].split("\n")
n = 0
while n < len(lines) and lines[n].startswith("|"):
    n += 1
b = a + len("\n".join(lines[:n]))
open(path, "w").write(s[:a] + "\n".join(rows) + s[b:])
print(len(rows) - 2, "rows")
