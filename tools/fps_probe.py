"""FPS probe: ms for m keypoints of two clouds of n points (one launch, as the pair path runs it)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bufferx_amd as bx
from bufferx_amd import lib
m = 5000
cfg = bx.make_cfg("3DMatch"); cfg.patch.num_fps = m
pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
for n in [int(a) for a in sys.argv[1:]] or [38000]:
    pair = bx.synth.make_pair(100, os.environ.get("KIND", "indoor"), n_target=n, shared=os.environ.get("KIND", "indoor") == "indoor")
    ctx = lib.Context(cfg, max_points=max(len(pair["src"]), len(pair["tgt"])), device=0, packed_weights=pw)
    S = cfg.patch.num_scales
    ns, nt = len(pair["src"]), len(pair["tgt"])
    rng = np.random.default_rng(0)
    dev = torch.device("cuda:0")
    ps = torch.from_numpy(np.stack([rng.permutation(ns) for _ in range(S)]).astype(np.int32)).to(dev)
    pt = torch.from_numpy(np.stack([rng.permutation(nt) for _ in range(S)]).astype(np.int32)).to(dev)
    src, tgt = torch.from_numpy(pair["src"]).to(dev), torch.from_numpy(pair["tgt"]).to(dev)
    st = torch.cuda.current_stream()
    for it in range(3):
        ctx.register_pair_async(src, tgt, pair["aligned_z"], ps, pt, 5); st.synchronize()
    ctx.profile_enable(True)
    for it in range(3):
        ctx.register_pair_async(src, tgt, pair["aligned_z"], ps, pt, 5); st.synchronize()
    pr = ctx.profile_read()
    print("n", ns, nt, "PPT", os.environ.get("BX_FPS_PPT", "auto"), "fps ms %.3f" % (pr["fps"][0] / pr["fps"][1]), "us/sample %.3f" % (pr["fps"][0] / pr["fps"][1] / m * 1e3), flush=True)
    if os.environ.get("BX_FPS_TRACE"):
        import ctypes as C
        buf = (C.c_int64 * 64)()
        ctx.lib.bx_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
        rc = ctx.lib.bx_debug_read(ctx.handle, buf, 64)
        print("flags (fast + 2*colocate + 100*G):", buf[63])
        a = np.array(buf[:], np.int64).reshape(8, 8)
        print("round stamps (cycles): apply + entries | barrier 1 | publish K entries + bound | poll | resolve | barrier 2 ; samples resolved ; gap to the next round")
        for r in range(8):
            d = a[r, :7] - a[r, 0]
            gap = (a[r + 1, 0] - a[r, 6]) if r < 7 else 0
            print("  ", [int(d[i] - d[i - 1]) for i in range(1, 7)], "total", int(d[6]), "samples", int(a[r, 7]), "gap", int(gap))
    ctx.close()
