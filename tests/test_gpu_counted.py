"""The hit-count hand-over between the neighbour gather and the patch kernels (round 5): bx_ball_group_counted writes only the REAL
slots of a patch and their number, bx_patch_features_counted takes (patches, counts, keypoints) -- and both must reproduce the padded
form bit for bit: the reference pads a patch with its first hit and then replaces the padded slots and slot P - 1 by the keypoint
(models/patch_embedder.py:99-111), so the slots beyond the count ARE the keypoint, the zero vector after centring; SPT zeroes its own
padded samples (utils/common.py:440-447), which is why a zero point that fills a sample slot and an empty slot contribute the same
value.  The padded form itself is checked against the oracle in tests/test_gpu_stages.py; here the counted form == the padded form on
the same inputs (bx_register_pair runs the counted form, so tests/test_gpu_pipeline.py et al. check it end to end as well) -- and, since
round 6, the counted kernels against the ORACLE directly (count, real slots, R, features)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _check(ctx, pts_perm, kp, r, P, aligned_modes=(False, True), oracle=None, packed=None, oracle_rows=None):
    """counted form == padded GPU form, and (oracle given) counted form == the ORACLE directly: the real slots and the count against
    oracle.ball_group's padded index list / patches, R and the features against oracle.patch_features on the oracle's own padded
    patches -- so the whole-pair path's kernels do not rest on the padded GPU form as a go-between."""
    import torch
    from bufferx_amd import lib
    rad = torch.tensor([float(r)], dtype=torch.float64)
    idx, patches = ctx.ball_group(pts_perm, kp, rad, P)
    pc, cnt = ctx.ball_group_counted(pts_perm, kp, rad, P)
    idx, patches, pc_np, cnt = _np(idx), _np(patches), _np(pc), _np(cnt)
    if oracle is not None:
        ridx, rpatches = oracle.ball_group(np.asarray(pts_perm, np.float32), np.asarray(kp, np.float32), np.float32(r), P)
        rreal = 1 + (ridx[:, 1:] != ridx[:, :1]).sum(1)
        assert np.array_equal(cnt, np.minimum(rreal, P - 1)), "count vs oracle"
        rlive = np.arange(P)[None, :] < cnt[:, None]
        assert np.array_equal(pc_np[rlive], rpatches[rlive]), "real slots vs oracle"
        kpo = np.broadcast_to(np.asarray(kp, np.float32)[:, None, :], rpatches.shape)
        assert np.array_equal(rpatches[~rlive], kpo[~rlive]), "the oracle's padded slots are the keypoint"
        rows = slice(None) if oracle_rows is None else oracle_rows         # (the oracle's SPT is a scalar CPU loop: sample at real size)
        for aligned in aligned_modes:
            R1, f1 = ctx.patch_features_counted(pc, torch.from_numpy(cnt), kp, rad, aligned)
            rR, rfeat = oracle.patch_features(rpatches[rows], r, aligned, packed["pnt_w"], packed["pnt_b"])
            assert np.array_equal(_np(R1)[rows], rR), ("R vs oracle", aligned)
            assert np.array_equal(lib.chunked_to_logical(_np(f1)[rows]), rfeat), ("features vs oracle", aligned)
    # the count: hits clamped to [1, P - 1]; the padded index list repeats its first entry beyond the hits
    real = 1 + (idx[:, 1:] != idx[:, :1]).sum(1)
    assert np.array_equal(cnt, np.minimum(real, P - 1))
    j = np.arange(P)[None, :]
    live = j < cnt[:, None]
    assert np.array_equal(pc_np[live], patches[live])                         # the real slots, bit for bit
    assert np.isnan(pc_np[~live]).all()                                       # nothing else is written
    kpb = np.broadcast_to(np.asarray(kp, np.float32)[:, None, :], patches.shape)
    assert np.array_equal(patches[~live], kpb[~live])                         # and what is not written IS the keypoint in the padded form
    out = {}
    for aligned in aligned_modes:
        R0, f0 = ctx.patch_features(patches, rad, aligned)
        R1, f1 = ctx.patch_features_counted(pc, cnt, kp, rad, aligned)
        assert np.array_equal(_np(R0), _np(R1)), ("R", aligned)
        assert np.array_equal(_np(f0), _np(f1)), ("feat", aligned)
        out[aligned] = (int(cnt.min()), int(cnt.max()), float(cnt.mean()))
    return out


@pytest.fixture(scope="module")
def ctx(bx, packed):
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 256, 128, 1
    cfg.patch.search_radius_thresholds = [5]
    cfg.patch.num_points_radius_estimate = 256
    c = lib.Context(cfg, max_points=70000, device=0, packed_weights=packed)
    yield c
    c.close()


@pytest.mark.parametrize("n,K,P,r", [(5000, 200, 64, 0.35), (5000, 200, 128, 0.2), (30000, 256, 128, 0.12), (30000, 256, 128, 0.04), (2500, 100, 96, 2.0)])
def test_counted_equals_padded(ctx, oracle, bx, packed, n, K, P, r):
    """full balls (count = P - 1), partly filled ones, nearly empty ones (r = 0.04: most keypoints hold a handful of points);
    against the padded GPU form AND against the oracle directly"""
    pts = bx.synth.make_pair(n, "indoor", n_target=n)["src"]
    pp = pts[oracle.make_perm(len(pts), 5, 0)]
    kp = pts[oracle.fps(pts, K)]
    print(_check(ctx, pp, kp, r, P, oracle=oracle, packed=packed))


def test_counted_empty_balls_and_point_zero(ctx, oracle, packed):
    """keypoints without a single hit (slot 0 = point 0 of the permuted cloud, the reference's quirk: count 1), keypoints whose first
    hit IS point 0 (sphere_query's `group_idx[:, :, 0] == 0` mask) and exact duplicates of cloud points"""
    rng = np.random.default_rng(11)
    pts = rng.random((4000, 3), np.float32)
    kp = np.concatenate([rng.random((24, 3), np.float32) * 3 + 2,           # far away: no hits
                         pts[:1].repeat(4, 0) + np.float32([[0, 0, 0], [0.01, 0, 0], [0, 0.02, 0], [0.05, 0.05, 0]]),   # point 0 inside the ball
                         pts[100:164]]).astype(np.float32)
    _check(ctx, pts, kp, 0.08, 64, oracle=oracle, packed=packed)
    _check(ctx, pts, kp, 0.3, 64, oracle=oracle, packed=packed)


def test_counted_at_real_size(bx, packed, oracle):
    """K = 5000 / P = 1024 at the three scales' typical radii on a 45k-point fragment (the whole-pair path's shapes)"""
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch = 5000, 1024
    c = lib.Context(cfg, max_points=70000, device=0, packed_weights=packed)
    try:
        pts = bx.synth.make_pair(7, "indoor", n_target=45000, shared=True)["src"]
        pp = pts[oracle.make_perm(len(pts), 3, 0)]
        kp = pts[oracle.fps(pts, 5000)]
        for r in (0.55, 0.33, 0.16):
            print("r", r, _check(c, pp, kp, r, 1024, aligned_modes=(False,), oracle=oracle, packed=packed, oracle_rows=slice(0, 5000, 125)))
    finally:
        c.close()


def test_counted_counts_are_clamped(ctx, oracle, bx):
    """counts[] is device memory handed through a public C-ABI entry: values outside [1, P - 1] are clamped into it by the kernels,
    never used as an LDS index (include/bufferx.h; round-5 advisor finding)."""
    import torch
    P = 64
    pts = bx.synth.make_pair(3, "indoor", n_target=4000)["src"]
    kp = pts[oracle.fps(pts, 64)]
    rad = torch.tensor([0.3], dtype=torch.float64)
    pc, cnt = ctx.ball_group_counted(pts, kp, rad, P, fill=0.0)
    good = _np(cnt).copy()
    bad = good.copy()
    bad[0::4], bad[1::4], bad[2::4] = 0, -7, P + 1000        # every fourth count stays as it is
    want = np.clip(bad, 1, P - 1)
    for aligned in (False, True):
        R0, f0 = ctx.patch_features_counted(pc, torch.from_numpy(want.astype(np.int32)), kp, rad, aligned)
        R1, f1 = ctx.patch_features_counted(pc, torch.from_numpy(bad.astype(np.int32)), kp, rad, aligned)
        assert np.array_equal(_np(R0), _np(R1)) and np.array_equal(_np(f0), _np(f1))
