"""Latency-form probe: per-stage event times of a tiled context (tag 0 = the FPS launches on the context's own stream)."""
import copy, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bufferx_amd as bx
from bufferx_amd import lib

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = bx.make_cfg("3DMatch")
cfg.match.enable_early_exit = False
cfg.test.keypoint_tiles = tiles
cfg.patch.num_fps = 5000
pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
pair = bx.synth.make_pair(100, "indoor", n_target=40000, shared=True)
S = cfg.patch.num_scales
rng = np.random.default_rng(0)
ns, nt = len(pair["src"]), len(pair["tgt"])
dev = torch.device("cuda:0")
ps = torch.from_numpy(np.stack([rng.permutation(ns) for _ in range(S)]).astype(np.int32)).to(dev)
pt = torch.from_numpy(np.stack([rng.permutation(nt) for _ in range(S)]).astype(np.int32)).to(dev)
src, tgt = torch.from_numpy(pair["src"]).to(dev), torch.from_numpy(pair["tgt"]).to(dev)
ctx = lib.Context(cfg, max_points=max(ns, nt), device=0, packed_weights=pw)
st = torch.cuda.current_stream()
for it in range(8):
    if it == 4:
        ctx.profile_enable(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    r = ctx.register_pair_async(src, tgt, pair["aligned_z"], ps, pt, 5)
    b.record(st)
    st.synchronize()
    print("tiles", tiles, "iter", it, "ms %.3f" % a.elapsed_time(b), "n", ns, nt, flush=True)
pr = ctx.profile_read()
print({k: (round(v[0] / 4, 3), v[1] // 4) for k, v in pr.items() if v[1]})
ctx.close()
