"""No register spills / scratch in the decode path's kernels (not gpu: reads the built library's code-object metadata).

Found in round 6 by a PMC pass (profiles/r06_pmc_dattn.txt: 4.5 MB of WRITE_SIZE per launch of the 3-bit fused attention kernel): the
compiler had parked a quad of early-requested o_proj words in scratch -- a wait for the load right where it is issued, and 5 MB of
scratch traffic per launch in the CU's in-order memory queue.  scratch/kernel_resources.py lists every kernel of a library."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gbx-lm_amd", "csrc", "libgbx_qlinear.so")

# the launches of FusedDecodeRunner / DecodeRunner at the benched shapes (8B, 3.2-3B; 2 / 3 / 4 bits)
HOT = re.compile(r"gbx::(dblock_kernel|dmlp_kernel|dmlpv_kernel|dattn_kernel|dgemv_mx_kernel|dense_gemv_kernel|attn_decode\w*kernel|"
                 r"embed_next_kernel|acc_finish_kernel|slices_norm_xcd_kernel)<")
# known and accepted: the 2-bit down_proj of the five- / three-launch block (two lane passes per row; the one- and two-launch blocks
# that replaced it do not use it), 2-4 registers
KNOWN = re.compile(r"gbx::dgemv_kernel<2, 2, [14], 16, 2, [01], 2, false, (true|false)>")


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_decode_kernels_have_no_spills():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "kernel_resources.py"), LIB, "--all"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " vgpr " in l]
    assert len(lines) > 500, "metadata of the library's kernels not found"
    hot = [l for l in lines if HOT.search(l)]
    assert len(hot) >= 20, hot
    bad = [l for l in hot if not re.search(r"vgpr_spill\s+0 ", l) or not re.search(r"scratch\s+0 B", l)]
    assert not bad, "spills / scratch in decode-path kernels:\n" + "\n".join(bad)
    dg = [l for l in lines if "gbx::dgemv_kernel<" in l and (not re.search(r"vgpr_spill\s+0 ", l) or not re.search(r"scratch\s+0 B", l))]
    unknown = [l for l in dg if not KNOWN.search(l)]
    assert not unknown, "new spills in the decode GEMV family:\n" + "\n".join(unknown)
