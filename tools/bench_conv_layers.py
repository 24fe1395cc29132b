#!/usr/bin/env python
"""tools/bench_conv_layers.py -- per-layer time of the Cylindrical_Net stack at K units through bx_conv_layer (hipEvents on the
kernels' stream, pre-allocated maps), one JSON line.  With BX_W43_STAMPS=1 (a library built with -DBX_W43_STAMP=1) also the phase
stamps of workgroup (0, 0).  Checks nothing against the oracle (the -m gpu tests do)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(3, 64), (4, 64), (4, 128), (8, 128), (8, 64), (4, 64), (4, 32), (2, 32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=5000)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tag", default="")
    ap.add_argument("--form", default=None, help="cfg.arith.desc_conv (default: the library default)")
    args = ap.parse_args()
    import torch
    import bufferx_amd as bx
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = args.K, 64, 1
    cfg.patch.search_radius_thresholds = [5]
    if args.form:
        cfg.arith.desc_conv = args.form
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    ctx = lib.Context(cfg, max_points=4096, device=0, packed_weights=pw)
    K = args.K
    g = torch.Generator(device="cuda:0").manual_seed(1)
    bufs = [torch.randn((K, 8, 140, 16), device="cuda:0", generator=g).abs_() for _ in range(2)]
    st = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    res = {}
    for l, (nch, cout) in enumerate(SHAPES):
        x = bufs[l & 1][:, :nch].contiguous()
        y = torch.empty((K, (cout + 15) // 16, 140, 16), device="cuda:0")

        def run():
            rc = ctx.lib.bx_conv_layer(ctx.handle, st, C.c_int32(0), C.c_int32(l), C.c_void_p(x.data_ptr()), C.c_int32(K), C.c_void_p(y.data_ptr()))
            assert rc == 0, ctx.lib.bx_last_error()
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            run()
        b.record()
        torch.cuda.synchronize()
        res["L%d_%dx%d" % (l, nch, cout)] = round(a.elapsed_time(b) / args.iters * 1e3, 1)
    res["stack_us"] = round(sum(res.values()), 1)
    res["wino_us"] = round(sum(v for k, v in res.items() if k[:2] in ("L0", "L1", "L2", "L3", "L4", "L5")), 1)
    res["tag"] = args.tag
    res["form"] = args.form or "default"
    if os.environ.get("BX_W43_STAMPS"):
        buf = (C.c_int64 * 512)()
        ctx.lib.bx_debug_read(ctx.handle, buf, 512)
        a = np.array(buf[:]).reshape(32, 16)
        names = ["mfma", "transform", "barrierA", "vstore", "barrierB", "wreq", "groups", "output"]
        for l in range(6):
            for w, off in ((0, 0), (5, 8)):
                r = a[l, off:off + 8].astype(np.float64)
                groups = max(r[6], 1.0)
                chunks = groups * SHAPES[l][0]
                d = {n: int(r[i] / (groups if n == "output" else chunks)) for i, n in enumerate(names) if n != "groups"}
                d["groups"] = int(r[6])
                res["stamps_L%d_wave%d" % (l, w)] = d
    print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
