"""Per-kernel MFMA-pipe utilisation and effective clock from a rocprofv3 --pmc pass (rocpd sqlite).
usage: python tools/pmc_mfma.py <dir with *.db> [name filter]
  util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)      (matrix-pipe busy share of the elapsed shader cycles)
  clock  = GRBM_GUI_ACTIVE / duration                                     (effective shader clock during the kernel, GHz)"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    i = n.find("(")
    return n[:i] if i > 0 else n


for f in sorted(glob.glob(os.path.join(root, "*.db"))):
    db = sqlite3.connect(f)
    rows = {}
    q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
    for name, ctr, n, v, d in db.execute(q):
        rows.setdefault(short(name), {})[ctr] = (n, v, d)
    print("== %s" % f)
    print("%-44s %5s %10s %8s %8s  %s" % ("kernel", "n", "avg_us", "GHz", "mfma%", "other counters (mean per dispatch)"))
    for k, c in sorted(rows.items(), key=lambda kv: -(kv[1].get("GRBM_GUI_ACTIVE", (0, 0, 0))[1] or 0)):
        if flt and flt not in k:
            continue
        g = c.get("GRBM_GUI_ACTIVE")
        m = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if not g:
            continue
        dur_us = (g[2] or 0) / 1e3
        ghz = g[1] / (g[2] or 1)
        util = (m[1] / (g[1] * 1024.0) * 100.0) if m else float("nan")
        rest = " ".join("%s=%.3g" % (n, v[1]) for n, v in sorted(c.items()) if n not in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"))
        print("%-44s %5d %10.1f %8.3f %8.1f  %s" % (k[:44], g[0], dur_us, ghz, util, rest))
