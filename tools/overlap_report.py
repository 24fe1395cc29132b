"""How the kernels of several pairs in flight share the GPU: from a rocprofv3 --kernel-trace database of `bench.py` (8 pairs in
flight) compute, over the steady-state half of the run, the wall time, the union of all kernel intervals (GPU not idle), the sum
of kernel durations (serial sum) and the time-weighted number of kernels running at once; split MFMA convolution kernels / FPS / rest.
usage: python tools/overlap_report.py <dir with *_results.db> <pairs in the window are estimated from state_reset_kernel launches>"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
f = sorted(glob.glob(os.path.join(root, "*.db")))[0]
db = sqlite3.connect(f)
rows = list(db.execute("select name, start, end from kernels order by start"))
grid = {}
for n, gx, gy, gz, wx in db.execute("select name, grid_x, grid_y, grid_z, workgroup_x from kernels"):
    grid.setdefault(n, []).append(gx * gy * gz // max(wx, 1))
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo, hi = t0 + (t1 - t0) * 0.35, t0 + (t1 - t0) * 0.75          # steady state: inside the timed region of the bench
win = [(n, max(s, lo), min(e, hi)) for n, s, e in rows if e > lo and s < hi]
wall = hi - lo


def cls(n):
    if "conv_kernel" in n or "conv32_kernel" in n or "cost_l1" in n or "wino" in n:
        return "mfma"
    if "fps_kernel" in n:
        return "fps"
    return "other"


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


pairs = sum(1 for n, s, e in rows if "state_reset_kernel" in n and lo <= s < hi)
ssum = {k: 0 for k in ("mfma", "fps", "other")}
for n, s, e in win:
    ssum[cls(n)] += e - s
ev = []
for n, s, e in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, lo, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
hist[depth] = hist.get(depth, 0) + (hi - last)
ms = 1e-6
print("window %.1f ms of the run (35 %% .. 75 %% of the trace), %d pairs started in it -> %.2f ms per pair" % (wall * ms, pairs, wall * ms / max(pairs, 1)))
print("GPU not idle (union of all kernel intervals): %.1f ms = %.1f %% of the window" % (union([(s, e) for _, s, e in win]) * ms, 100.0 * union([(s, e) for _, s, e in win]) / wall))
print("an MFMA kernel (conv / cost_l1) running:        %.1f ms = %.1f %%" % (union([(s, e) for n, s, e in win if cls(n) == "mfma"]) * ms, 100.0 * union([(s, e) for n, s, e in win if cls(n) == "mfma"]) / wall))
tot = sum(ssum.values())
print("serial sum of kernel durations: %.1f ms = %.2fx the window  (MFMA %.1f, FPS %.1f, other %.1f ms)" % (tot * ms, tot / wall, ssum["mfma"] * ms, ssum["fps"] * ms, ssum["other"] * ms))
print("per pair: serial sum %.2f ms (MFMA %.2f, FPS %.2f, other %.2f) against %.2f ms of wall" % tuple(x * ms / max(pairs, 1) for x in (tot, ssum["mfma"], ssum["fps"], ssum["other"], wall)))
print("kernels running at once (share of the window): " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / wall) for k, v in sorted(hist.items()) if v / wall > 0.002))

# per kernel inside the window: launches per pair, mean duration, mean workgroups per launch (to compare with the one-pair-in-flight profile)
per = {}
for n, s_, e in win:
    d = per.setdefault(n, [0, 0])
    d[0] += 1; d[1] += e - s_
print("%-58s %9s %10s %10s %9s" % ("kernel (window)", "per pair", "mean us", "ms / pair", "mean WGs"))
for n, (cnt, tot_) in sorted(per.items(), key=lambda kv: -kv[1][1])[:18]:
    g = grid.get(n, [0])
    print("%-58s %9.1f %10.1f %10.3f %9.0f" % (n[:58], cnt / max(pairs, 1), tot_ / cnt * 1e-3, tot_ * ms / max(pairs, 1), sum(g) / len(g)))
