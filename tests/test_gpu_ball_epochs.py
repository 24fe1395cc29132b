"""Neighbour gather with INDEX EPOCHS (round 4, k_ball.hip): the sweep of a keypoint's candidates ends behind the first index epoch
after which num_points_per_patch hits are known.  The stage entry point bx_ball_group always uses one epoch (it has no density hint),
so the epoch path is exercised through bx_register_pair with the capture armed: the captured select_patches output (reference
models/patch_embedder.py:92-120) of EVERY keypoint must equal the oracle's, bit for bit, in configurations whose expected ball
population (threshold % x cloud size) puts the set on 4 and on 8 epochs -- incl. keypoints whose balls hold fewer than P points (every
epoch swept), ball populations around P, and duplicate points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("n,P,thr,epochs", [(6000, 64, 5, 8), (6000, 160, 5, 4), (20000, 256, 5, 8), (20000, 512, 5, 4), (9000, 160, 2, 1)])
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
    for m in (len(src), len(tgt)):                              # the rule of bxk_ball_grids puts both clouds on `epochs` epochs
        hits = thr * 0.01 * m
        assert (8 if hits >= 3.0 * P else (4 if hits >= 1.4 * P else 1)) == epochs, (m, hits)
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
                assert cnt.max() > P and cnt.min() < cnt.max()      # balls above P exist (the sweep ends early for them)
        ctx.set_capture(None, 0, 0)
    finally:
        ctx.close()
