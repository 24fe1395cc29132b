#!/usr/bin/env python
"""tools/pair_probe.py WORKLOAD [N] -- per-pair outcome of the synthetic benchmark pairs (seed, points, counts, RRE / RTE, registered)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    wl = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    import torch
    import bench
    import bufferx_amd as bx
    from bufferx_amd import lib
    cfg = bx.make_cfg(bench.WORKLOADS[wl][0])
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 5000, 1024, 3
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    ctx = lib.Context(cfg, max_points=130000, device=0, packed_weights=pw)
    for i in range(n):
        seed = 100 + i
        p = bench.make_pair(bx, wl, seed)
        rng = np.random.default_rng(seed)
        ps = np.stack([rng.permutation(len(p["src"])).astype(np.int32) for _ in range(3)])
        pt = np.stack([rng.permutation(len(p["tgt"])).astype(np.int32) for _ in range(3)])
        r = ctx.register_pair(p["src"], p["tgt"], p["aligned_z"], ps, pt, seed)
        pose = np.array(r.pose).reshape(4, 4)
        rre, rte = bx.synth.pose_error(pose, p["T_gt"])
        print(json.dumps(dict(seed=seed, n=[len(p["src"]), len(p["tgt"])], M=r.num_mutual, C=r.num_inlier_ind, inl=r.num_inliers, it=r.ransac_iters,
                              rre=round(float(rre), 3), rte=round(float(rte), 4), ok=bool(rre < cfg.test.rre_thresh and rte < cfg.test.rte_thresh))))
    ctx.close()


if __name__ == "__main__":
    main()
