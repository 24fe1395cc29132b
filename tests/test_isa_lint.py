"""The compiled F(4x4) kernels, checked without a GPU (tools/isa_lint.py cross-compiles k_wino43.hip / k_wino43v.hip to gfx950 assembly with
the flags of the shipped build): no VALU write to the data registers of a 16-byte store in the next issue slot -- the hazard that made
wino43v_kernel<6, 3, 64, 18, 2> store wrong values in round 5 (hipcc leaves no wait state behind a buffer_store with an SGPR offset) --, no
scratch reload between the stores of an output round and no vmcnt(0) drain inside the plane loop of the 64-column Cylindrical_Net kernels."""
import ast
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_isa_lint_clean():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py")], capture_output=True, text=True, timeout=1200)
    rows = [ast.literal_eval(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 19, out.stdout[-2000:] + out.stderr[-2000:]          # 7 Cylindrical_Net + 5 CostNet + 7 mixed-tile instantiations
    for r in rows:
        assert r["store_data_overwritten_next_slot"] == 0, r
        # (the mixed-tile kernel carries two specialised copies of its plane loop: 144 + 120 MFMAs)
        assert r["mfma"] == (264 if "wino43m" in r["kernel"] else 144) and r["vmcnt0_inside_plane_loop"] == 0, r
        assert r["scratch_in_plane_loop"] == 0, r
    # k_fps.hip: m0 (the lane select of the resolving wave's v_writelane block, inline asm) is used by that block only
    m0 = [l for l in out.stdout.splitlines() if l.startswith("k_fps_m0")]
    assert len(m0) == 1 and " foreign=0 " in m0[0] + " " and m0[0].endswith("detached=0") and " loads=3 " in m0[0], m0      # one block per points-per-thread instantiation
    assert out.returncode == 0, out.stdout[-2000:]
