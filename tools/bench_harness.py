#!/usr/bin/env python
"""tools/bench_harness.py -- end-to-end rate of harness.Runner at BASELINE configs[1] sizes: raw scans on disk (binary PLY) ->
native prefetch -> GPU voxel-size analysis + down-sampling + shuffle -> 3 pairs in flight -> evaluation rows.
Prints one JSON line; beside bench.py's number (inputs resident in HBM) it shows what file IO + pre-processing cost."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bufferx_amd as bx
    from bufferx_amd import harness

    def write_ply(path, pts):                   # binary little-endian PLY, x y z float32 (what 3DMatch fragments look like)
        with open(path, "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                     "property float z\nend_header\n" % len(pts)).encode("ascii"))
            f.write(np.ascontiguousarray(pts, "<f4").tobytes())
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 5000, 1024, 3
    cfg.test.pose_refine = True
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    pairs, raw_n = [], []
    for i in range(n_pairs):
        p = bx.synth.make_pair(100 + i, "indoor", n_target=int(rng.integers(20000, 60000)))
        files = []
        for k, c in (("s", p["src"]), ("t", p["tgt"])):
            raw = np.concatenate([c + rng.normal(0, 0.003, c.shape) for _ in range(4)]).astype(np.float32)
            f = os.path.join(d, f"{k}{i}.ply")
            write_ply(f, raw)
            files.append(f); raw_n.append(len(raw))
        pairs.append(dict(src_path=files[0], tgt_path=files[1], relt_pose=p["T_gt"]))
    mode = sys.argv[2] if len(sys.argv) > 2 else "reference"
    inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    run = harness.Runner(cfg, pw, device=0, inflight=inflight, max_raw_points=max(raw_n), max_points=90000, rng=mode)
    np.random.seed(0)
    run.run(pairs[:4])                         # warm-up
    torch.cuda.synchronize()
    for k in run.timers:
        run.timers[k] = 0.0
    t0 = time.perf_counter()
    rows, poses = run.run(pairs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(metric="pairs/s incl. file read, H2D, voxel analysis, down-sampling, shuffle, host RNG", value=round(n_pairs / dt, 3),
                          ms_per_pair=round(dt / n_pairs * 1e3, 2), pairs=n_pairs, rng=mode, inflight=inflight, mean_raw_points=int(np.mean(raw_n)),
                          host_prep_ms_per_pair=round(float(np.mean(rows[:, 8])) * 1e3, 2),
                          gpu_latency_ms_per_pair=round(float(np.mean(rows[:, 9])) * 1e3, 2),
                          host_ms_per_pair={k: round(v / n_pairs * 1e3, 2) for k, v in run.timers.items()})))


if __name__ == "__main__":
    main()
