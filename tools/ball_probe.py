#!/usr/bin/env python
"""tools/ball_probe.py WORKLOAD [SET] -- in-kernel phase stamps of the neighbour-gather query kernel of one (cloud, scale) set inside a
whole pair (BX_BALL_DEBUG = set + 1; BX_BALL_EPOCHS=1 forces one index epoch), plus the stage times of the pair."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    wl = sys.argv[1]
    st = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    os.environ["BX_BALL_DEBUG"] = str(st + 1)
    import torch
    import bench
    import bufferx_amd as bx
    from bufferx_amd import lib
    cfg = bx.make_cfg(bench.WORKLOADS[wl][0])
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 5000, 1024, 3
    if wl == "tiers":
        cfg.match.enable_early_exit = False
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    p = bench.make_pair(bx, wl, 100)
    ctx = lib.Context(cfg, max_points=max(len(p["src"]), len(p["tgt"])), device=0, packed_weights=pw)
    rng = np.random.default_rng(0)
    ps = np.stack([rng.permutation(len(p["src"])).astype(np.int32) for _ in range(3)])
    pt = np.stack([rng.permutation(len(p["tgt"])).astype(np.int32) for _ in range(3)])
    for _ in range(2):
        ctx.register_pair(p["src"], p["tgt"], p["aligned_z"], ps, pt, 1)
    ctx.profile_enable(True)
    ctx.register_pair(p["src"], p["tgt"], p["aligned_z"], ps, pt, 1)
    torch.cuda.synchronize()
    prof = {k: round(v[0], 4) for k, v in ctx.profile_read().items() if "neigh" in k}
    buf = (C.c_int64 * 480)()
    ctx.lib.bx_debug_read(ctx.handle, buf, 480)
    a = np.array(buf[:]).reshape(60, 8)
    med = np.median(a[:, 1:7], 0).astype(int)
    out = {"workload": wl, "set": st, "epochs_env": os.environ.get("BX_BALL_EPOCHS", "default"), "n": [len(p["src"]), len(p["tgt"])],
           "stage_ms": prof, "median_cycles": dict(zip(["setup", "t2", "scan", "rank", "output", "drained"], med.tolist())),
           "p90_cycles": np.percentile(a[:, 1:7], 90, axis=0).astype(int).tolist(), "pieces_median": int(np.median(a[:, 7])), "pieces_max": int(a[:, 7].max())}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
