#!/usr/bin/env python
"""tools/bench_pre.py -- timing of the GPU pre-processing entry points (bx_pre_voxel_downsample, bx_pre_pca) on a raw-sized cloud,
(GPU time per call, hipEvent-timed).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bufferx_amd as bx
    from bufferx_amd import lib
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    vs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.025
    rng = np.random.default_rng(0)
    base = np.ascontiguousarray(bx.synth.make_pair(5, "indoor", n_target=30000)["src"], np.float32)
    pts = (base[rng.integers(0, len(base), n)] + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
    cfg = bx.make_cfg("3DMatch")
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    ctx = lib.Context(cfg, max_points=1024, device=0, packed_weights=pw)
    ctx.pre_reserve(n)
    d = torch.from_numpy(pts).to("cuda:0")
    idx = torch.from_numpy(rng.choice(n, n // 10, replace=False).astype(np.int32)).to("cuda:0")

    def t(fn, it=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it
    ms_vox = t(lambda: ctx.pre_voxel_downsample(d, vs))
    ms_pca = t(lambda: ctx.pre_pca(d, idx))
    m = int(ctx.pre_voxel_downsample(d, vs)[1].cpu().numpy()[0])
    print(json.dumps(dict(n=n, voxel=vs, voxels=m, gpu_voxel_ms=round(ms_vox, 3), gpu_pca_ms=round(ms_pca, 3))))
    ctx.close()


if __name__ == "__main__":
    main()
