"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel time table and per-kernel mean FETCH_SIZE / WRITE_SIZE.
usage: python tools/summarize_prof.py gpurun_out/prof_r01 > profiles/r01_summary.txt"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    i = n.find("(")
    return n[:i] if i > 0 else n


for f in sorted(glob.glob(os.path.join(root, "kt", "*.db"))):
    db = sqlite3.connect(f)
    print("== rocprofv3 --kernel-trace --stats :: per-kernel (", os.path.relpath(f, root), ")")
    print("%-52s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-52s %6d %12.3f %12.2f %7.2f" % (short(name)[:52], calls, tot / 1e3, avg, pct))
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in sorted(glob.glob(os.path.join(root, tag, "*.db"))):
        db = sqlite3.connect(f)
        print("\n== rocprofv3 --pmc %s :: mean per dispatch (counter unit: KiB; gfx950: FETCH_SIZE reads 1/2 of wide coalesced reads,"
              " MI355X_MICROARCH.md §HBM)" % ctr)
        q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=? "
             "group by kernel_name order by sum(value) desc limit 30")
        print("%-52s %6s %14s %12s" % ("kernel", "n", "mean_KiB", "avg_us"))
        for name, n, v, d in db.execute(q, (ctr,)):
            print("%-52s %6d %14.1f %12.2f" % (short(name)[:52], n, v, (d or 0) / 1e3))

for f in sorted(glob.glob(os.path.join(root, "pmc_mfma", "*.db"))):
    db = sqlite3.connect(f)
    print("\n== rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE :: per kernel (busy = matrix-pipe busy share of the elapsed shader"
          " cycles = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs); GHz = GUI_ACTIVE / 8 / duration)")
    rows = {}
    q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
    for name, ctr, n, v, d in db.execute(q):
        rows.setdefault(short(name), {})[ctr] = (n, v, d)
    print("%-52s %6s %12s %8s %8s" % ("kernel", "n", "avg_us", "GHz", "busy"))
    for k, c in sorted(rows.items(), key=lambda kv: -(kv[1].get("GRBM_GUI_ACTIVE", (0, 0, 0))[1] or 0) * (kv[1].get("GRBM_GUI_ACTIVE", (0, 0, 0))[0] or 0)):
        g, m = c.get("GRBM_GUI_ACTIVE"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if not g or not m or m[1] <= 0:
            continue
        cyc = g[1] / 8.0
        print("%-52s %6d %12.2f %8.3f %8.3f" % (k[:52], g[0], (g[2] or 0) / 1e3, cyc / (g[2] or 1), m[1] / (cyc * 1024.0)))
