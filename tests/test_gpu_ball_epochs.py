"""select_patches of EVERY keypoint through the whole-pair path (capture), across ball populations from far below to far above
num_points_per_patch -- the cases an index-epoch sweep was built and tested on in round 4 (profiles/r04_ball_epochs.txt: bit-exact,
15-17 % slower, removed again); kept as a parity test of the grid-accelerated neighbour gather inside bx_register_pair: the captured
patches (reference models/patch_embedder.py:92-120) must equal the oracle's bit for bit, incl. keypoints whose balls hold fewer than P
points, populations around and several times P (only the first P indices survive), and duplicate points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("n,P,thr,epochs",      # `epochs`: expected population / P is >= 3 (8), >= 1.4 (4), below (1)
                         [(6000, 64, 5, 8), (6000, 160, 5, 4), (20000, 256, 5, 8), (20000, 512, 5, 4), (9000, 160, 2, 1)])
def test_captured_patches_equal_oracle(bx, packed, oracle, n, P, thr, epochs):
    import torch
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    K = 300
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = K, P, 1
    cfg.patch.search_radius_thresholds = [thr]
    cfg.patch.num_points_radius_estimate = 256
    cfg.match.iter_n = 500
    pair = bx.synth.make_pair(31 + n % 7, "indoor", n_target=n, shared=True)
    src = pair["src"].copy()
    src[100:140] = src[50:90]                                   # duplicate points: equal distances, distinct indices
    tgt = pair["tgt"]
    ctx = lib.Context(cfg, max_points=max(len(src), len(tgt)), device=0, packed_weights=packed)
    try:
        rng = np.random.default_rng(5)
        ps = rng.permutation(len(src)).astype(np.int32)[None]
        pt = rng.permutation(len(tgt)).astype(np.int32)[None]
        for cloud, pts, perm in ((0, src, ps), (1, tgt, pt)):
            cap = ctx.set_capture(0, cloud, len(pts))
            r = ctx.register_pair(src, tgt, pair["aligned_z"], ps, pt, 3)
            torch.cuda.synchronize()
            assert r.status == 0
            kp = _np(cap["kpts"][cloud])
            pp = _np(cap["pts_perm"])[:len(pts)]
            assert np.array_equal(pp, pts[perm[0]])
            _, ref = oracle.ball_group(pp, kp, np.float32(r.des_r[0]), P)
            got = _np(cap["patches"])
            assert np.array_equal(got, ref), (cloud, int((got != ref).any(axis=(1, 2)).sum()))
            d2 = ((kp[:, None, :].astype(np.float64) - pp[None, :, :]) ** 2).sum(-1)
            cnt = (d2 < float(np.float32(r.des_r[0])) ** 2).sum(1)
            if epochs > 1:
                assert cnt.max() > P and cnt.min() < cnt.max()      # balls above P exist: only the first P indices survive
        ctx.set_capture(None, 0, 0)
    finally:
        ctx.close()
