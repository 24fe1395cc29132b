#!/usr/bin/env python
"""tools/bench_stage.py -- per-stage micro-benchmarks of the HIP hot path at BASELINE cfg-2 sizes (K=5000, P=1024).

  python tools/bench_stage.py ball [--n 30000] [--iters 20]
  python tools/bench_stage.py patch
  python tools/bench_stage.py conv

Times the stage through the C-ABI with HIP events on the stream the kernels run on; prints one JSON line per case.
Checks nothing against the oracle (the -m gpu tests do that)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(torch, fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["ball", "patch", "conv", "radius", "all"])
    ap.add_argument("--n", type=int, default=30000)
    ap.add_argument("--K", type=int, default=5000)
    ap.add_argument("--P", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kind", default="indoor")
    args = ap.parse_args()
    import torch
    import bufferx_amd as bx
    from bufferx_amd import lib
    K, P = args.K, args.P
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = K, P, 3
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    pair = bx.synth.make_pair(11, args.kind, n_target=args.n)
    pts = np.ascontiguousarray(pair["src"], np.float32)
    n = len(pts)
    ctx = lib.Context(cfg, max_points=max(n, 1024), device=0, packed_weights=pw)
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    perm = torch.from_numpy(rng.permutation(n).astype(np.int32)).to(dev)
    dpts = torch.from_numpy(pts).to(dev)
    idx, kp = ctx.fps(dpts, K)
    pp = ctx.permute(dpts, perm)
    radii = ctx.radius(dpts, n, kp[:2000].contiguous(), [5, 2, 0.5])
    torch.cuda.synchronize()
    rr = radii.cpu().numpy()
    out = []
    if args.what in ("radius", "all"):
        kp2 = kp[:2000].contiguous()
        us = timeit(torch, lambda: ctx.radius(dpts, n, kp2, [5, 2, 0.5]), args.iters)
        out.append(dict(stage="radius (memset + radius_hist_kernel + radius_bisect_kernel)", n=n, nk=2000, us=round(us, 1), slices=os.environ.get("BX_RAD_SLICES", "default"),
                        Gdist_per_s=round(2000.0 * n / us / 1e3, 1), radii=[float(v) for v in rr]))
    if args.what in ("ball", "all"):
        for si in range(3):
            rad = radii[si:si + 1].contiguous()
            for want_idx in (True, False):
                us = timeit(torch, lambda: ctx.ball_group(pp, kp, rad, P, want_idx=want_idx), args.iters)
                nbytes = 12.0 * n + 12.0 * K + (4.0 * K * P if want_idx else 0.0) + 12.0 * K * P
                if os.environ.get("BX_BALL_DEBUG") and want_idx:
                    import ctypes as C
                    buf = (C.c_int64 * 480)()
                    ctx.lib.bx_debug_read(ctx.handle, buf, 480)
                    a = np.array(buf[:]).reshape(60, 8)
                    t0 = a[:, 0] - a[:, 0].min()
                    print("phase cycles (median over 60 sampled waves) setup %d rows %d scan %d expand %d output %d drained %d | T med %d | start spread: med %d max %d"
                          % tuple(list(np.median(a[:, 1:7], 0).astype(int)) + [int(np.median(a[:, 7])), int(np.median(t0)), int(t0.max())]))
                    print("  p90:", np.percentile(a[:, 1:7], 90, axis=0).astype(int))
                out.append(dict(stage="ball_group(all kernels of the stage)", n=n, K=K, P=P, radius=float(rr[si]), idx=want_idx,
                                us=round(us, 2), GBps=round(nbytes / us / 1e3, 1), frac_of_8TBps=round(nbytes / us / 1e3 / 8000, 4)))
    if args.what in ("patch", "all"):
        for si in range(3):
            rad = radii[si:si + 1].contiguous()
            _, patches = ctx.ball_group(pp, kp, rad, P, want_idx=False)
            for aligned in (False, True):
                us = timeit(torch, lambda: ctx.patch_features(patches, rad, aligned), args.iters)
                if os.environ.get("BX_BALL_DEBUG"):
                    import ctypes as C
                    buf = (C.c_int64 * 480)()
                    ctx.lib.bx_debug_read(ctx.handle, buf, 480)
                    a = np.array(buf[:]).reshape(60, 8)
                    print("patch phases (median cycles, wave 0): normalised %d rowlists %d query %d conv+store %d"
                          % tuple(np.median(a[:, 1:5], 0).astype(int)))
                out.append(dict(stage="patch_features", K=K, P=P, radius=float(rr[si]), aligned=aligned, us=round(us, 2)))
    if args.what in ("conv", "all"):
        _, patches = ctx.ball_group(pp, kp, radii[0:1].contiguous(), P, want_idx=False)
        _, feat = ctx.patch_features(patches, radii[0:1].contiguous(), False)
        if os.environ.get("BX_BENCH_ZERO"):   # power experiment: all-zero A operands
            feat.zero_()
        us = timeit(torch, lambda: ctx.desc_net(feat), max(3, args.iters // 4))
        if os.environ.get("BX_BALL_DEBUG"):
            import ctypes as C
            buf = (C.c_int64 * 512)()
            ctx.lib.bx_debug_read(ctx.handle, buf, 512)
            a = np.array(buf[:]).reshape(16, 32)
            print("conv layer-1 stamps (median cycles over 16 workgroups): prologue %d | " % np.median(a[:, 1]) +
                  " ".join("c%d taps %d bar %d" % (c, np.median(a[:, 2 + 2 * c]), np.median(a[:, 3 + 2 * c])) for c in range(4)) + " | end %d" % np.median(a[:, 10]))
        flops = 2.0 * 59.351e6 * K
        out.append(dict(stage="desc_net (8 conv + head)", K=K, us=round(us, 1), TFLOPs=round(flops / us / 1e6, 2)))
    for o in out:
        print(json.dumps(o))
    ctx.close()


if __name__ == "__main__":
    main()
