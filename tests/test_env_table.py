"""The environment switches are rows of ONE table per side (gbx-lm_amd/env.py:TABLE, csrc/gbx_env.h:env_table); DESIGN.md section 8 documents
every row with its default.  (CPU: text only, nothing is loaded.)"""
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernel_rows():
    src = open(os.path.join(ROOT, "gbx-lm_amd", "csrc", "gbx_env.h")).read()
    return re.findall(r'\{"(GBX_[A-Z0-9_]+)",\s*(-?\d+),', src)


def _design_rows():
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = doc[doc.index("## 8. Environment switches"):]
    return {m.group(1): m.group(2).strip() for m in re.finditer(r"^\| `(GBX_[A-Z0-9_]+)` \| ([^|]*) \|", sec, re.M)}


def test_every_kernel_switch_is_documented_with_its_default():
    rows, doc = _kernel_rows(), _design_rows()
    assert len(rows) >= 20 and len({n for n, _ in rows}) == len(rows), "duplicate row in csrc/gbx_env.h"
    for name, default in rows:
        assert name in doc, f"{name} (csrc/gbx_env.h) has no row in DESIGN.md section 8"
        assert doc[name] == default, f"{name}: DESIGN.md says default {doc[name]!r}, csrc/gbx_env.h says {default}"


def test_every_host_switch_is_documented():
    env = importlib.import_module("gbx_lm_amd.env")
    doc = _design_rows()
    for name in env.TABLE:
        assert name in doc, f"{name} (gbx-lm_amd/env.py) has no row in DESIGN.md section 8"


def test_no_stale_rows_in_the_document():
    env = importlib.import_module("gbx_lm_amd.env")
    known = {n for n, _ in _kernel_rows()} | set(env.TABLE)
    stale = sorted(set(_design_rows()) - known)
    assert not stale, f"DESIGN.md section 8 documents switches no table knows: {stale}"


def test_no_stray_switch_reads():
    """Every GBX_* environment read goes through the two table readers (csrc/gbx_env.h:env_switch, gbx-lm_amd/env.py:get): a raw
    getenv / os.environ read of a GBX_ name outside them is either one of the documented diagnostic-build / experimental-library
    names or a bug (a switch without a row of documentation, or one read with different truthiness in two places)."""
    import glob
    allowed_raw = {"GBX_STAMP_PTR", "GBX_STAMP_SLOTS", "GBX_PROBE_SRC",       # -DGBX_STAMPS diagnostic builds (DESIGN.md section 8 preamble)
                   "GBX_GEMV_MX", "GBX_GEMV_MXD", "GBX_MXD_DEEP", "GBX_CHAIN_FLOOR"}   # libgbx_experimental.so's own opt-ins (gemv_mx.hip, gemv_mxd.hip)
    for f in glob.glob(os.path.join(ROOT, "gbx-lm_amd", "csrc", "*.h*")):
        if f.endswith("gbx_env.h"):
            continue
        for name in re.findall(r'getenv\("(GBX_[A-Z0-9_]+)"\)', open(f).read()):
            assert name in allowed_raw, f"{os.path.basename(f)} reads {name} with a raw getenv: route it through gbx::env_switch"
            if name in ("GBX_GEMV_MX", "GBX_GEMV_MXD", "GBX_MXD_DEEP", "GBX_CHAIN_FLOOR"):
                assert os.path.basename(f) in ("gemv_mx.hip", "gemv_mxd.hip"), f"{name} read outside the experimental kernels: {f}"
    py = [os.path.join(ROOT, "bench.py")] + glob.glob(os.path.join(ROOT, "gbx-lm_amd", "*.py")) + glob.glob(os.path.join(ROOT, "gbx-lm_amd", "models", "*.py"))
    for f in py:
        if f.endswith(os.path.join("gbx-lm_amd", "env.py")):
            continue
        src = open(f).read()
        hits = re.findall(r'os\.environ(?:\.get)?\(?\[?\s*["\'](GBX_[A-Z0-9_]+)', src)
        assert not hits, f"{os.path.relpath(f, ROOT)} reads {hits} from os.environ: use gbx_lm_amd.env.get / flag"
