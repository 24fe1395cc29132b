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
