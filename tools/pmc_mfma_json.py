"""MFMA-pipe occupancy and effective clock per kernel from a rocprofv3 --pmc pass (rocpd sqlite) -> JSON for bench.py's `mfma_busy`.
usage: python tools/pmc_mfma_json.py <dir with *.db>  > profiles/rNN_mfma_busy.json
  busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   share of the elapsed shader cycles in which a SIMD's
          matrix pipe was executing (the counter sums the SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' cycle counts)
  clock = GRBM_GUI_ACTIVE / 8 / duration                                        effective shader clock while the kernel ran (GHz)
Stack figures are weighted by kernel time (layers 1 and 5 of Cylindrical_Net share one instantiation: it runs twice per stack)."""
import glob
import json
import os
import sqlite3
import sys

root = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(root, "*.db"))):
    db = sqlite3.connect(f)
    q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
    for name, ctr, n, v, d in db.execute(q):
        k = name.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k[:k.find("(")] if "(" in k else k
        rows.setdefault(k, {})[ctr] = (n, v, d)

out = {"source": os.path.relpath(root) if os.path.isabs(root) else root, "units": "busy = matrix-pipe busy share of elapsed shader cycles; clock_GHz = effective shader clock", "kernels": {}}
for k, c in rows.items():
    g, m = c.get("GRBM_GUI_ACTIVE"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if not g or not m or m[1] <= 0:
        continue
    cyc = g[1] / 8.0
    out["kernels"][k] = {"dispatches": g[0], "avg_us": round((g[2] or 0) / 1e3, 2), "clock_GHz": round(cyc / (g[2] or 1), 3),
                         "busy": round(m[1] / (cyc * 1024.0), 4)}


def stack(pred):
    t = b = ck = 0.0
    for k, v in out["kernels"].items():
        if pred(k):
            w = v["avg_us"] * v["dispatches"]
            t += w; b += w * v["busy"]; ck += w * v["clock_GHz"]
    return (round(b / t, 4), round(ck / t, 3)) if t else (None, None)


d_busy, d_clk = stack(lambda k: k.startswith("conv_kernel<") and ", 140, 198, 140," in k or k.startswith("conv32_kernel<") or k.startswith("wino_kernel<") or k.startswith("wino_pair_kernel<") or k.startswith("wino43_kernel<"))
c_busy, c_clk = stack(lambda k: k.startswith("cost_l1_kernel") or k.startswith("wino_pose_kernel<") or k.startswith("wino43v_kernel<") or (k.startswith("conv_kernel<") and ", 140, 198, 140," not in k))   # MFMA kernels of CostNet (layers 1..9; the collapsed layer 0, cost_l0_kernel, is binary64 VALU work and has no MFMA cycles)
out["desc_conv_stack"] = d_busy
out["desc_conv_stack_clock_GHz"] = d_clk
out["costnet"] = c_busy
out["costnet_clock_GHz"] = c_clk
print(json.dumps(out, indent=1))
