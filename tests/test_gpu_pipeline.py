"""GPU end-to-end parity: bx_register_pair (C-ABI) vs the oracle pipeline and vs the committed golden fixtures
minted from the real reference code (tests/golden/make_golden.py)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    "indoor_small": ("3DMatch", "indoor", 3500, 11, False,
                     dict(num_fps=96, num_points_per_patch=96, num_scales=2, search_radius_thresholds=[5, 2],
                          num_points_radius_estimate=256), dict(iter_n=4000)),
    "indoor_success": ("3DMatch", "indoor", 5000, 3, True,
                       dict(num_fps=256, num_points_per_patch=128, num_scales=2, search_radius_thresholds=[2, 1],
                            num_points_radius_estimate=256), dict(iter_n=4000)),
    "indoor_early": ("3DMatch", "indoor", 5000, 3, True,
                     dict(num_fps=256, num_points_per_patch=128, num_scales=2, search_radius_thresholds=[2, 1],
                          num_points_radius_estimate=256), dict(iter_n=4000, enable_early_exit=True, early_exit_min_inliers=10)),
    "outdoor_small": ("KITTI", "outdoor", 0, 5, False,
                      dict(num_fps=80, num_points_per_patch=64, num_scales=1, search_radius_thresholds=[2],
                           num_points_radius_estimate=200), dict(iter_n=1200)),
    # outdoor configuration (aligned z, confidence 1.0, binary64 un-refined pose) over 3 scales
    "outdoor_3scale": ("KITTI", "outdoor", 0, 9, False,
                       dict(num_fps=128, num_points_per_patch=64, num_scales=3, search_radius_thresholds=[5, 2, 0.5],
                            num_points_radius_estimate=200), dict(iter_n=1200)),
    # 3 scales, early exit armed but never taken (two pose-estimation calls)
    "indoor_3scale": ("3DMatch", "indoor", 8000, 17, True,
                      dict(num_fps=256, num_points_per_patch=128, num_scales=3, search_radius_thresholds=[5, 2, 0.5],
                           num_points_radius_estimate=256), dict(iter_n=4000, enable_early_exit=True, early_exit_min_inliers=10 ** 6)),
    # BASELINE configs[0] (1 scale, 512 keypoints, 512 points per patch, RANSAC + refinement), minted by the reference's own forward
    "baseline_cfg0": ("3DMatch", "indoor", 20000, 21, True,
                      dict(num_fps=512, num_points_per_patch=512, num_scales=1, search_radius_thresholds=[5]), dict(iter_n=4000)),
    # round 4: reference-minted pairs at 1 000 - 1 500 keypoints x 3 scales (every other knob the dataset's default), incl. a 30 %-overlap
    # pair (not registered: seven consensus members) and an outdoor one with all 8 000 RANSAC iterations: matching is no longer sparse, consensus sets of a handful of members
    "mid_shared_a": ("3DMatch", "shared", 5000, 201, False, dict(num_fps=1200, num_points_per_patch=256, num_points_radius_estimate=600), dict()),
    "mid_shared_b": ("3DMatch", "shared", 6000, 202, False, dict(num_fps=1500, num_points_per_patch=256, num_points_radius_estimate=600), dict()),
    "mid_shared_c": ("3DMatch", "shared", 4000, 203, False, dict(num_fps=1000, num_points_per_patch=512, num_points_radius_estimate=500), dict()),
    "mid_low_overlap": ("3DLoMatch", "shared_low", 5000, 204, False, dict(num_fps=1500, num_points_per_patch=256, num_points_radius_estimate=600), dict()),
    "mid_kitti": ("KITTI", "outdoor_mid", 0, 205, False, dict(num_fps=1000, num_points_per_patch=256, num_points_radius_estimate=500), dict(iter_n=8000)),
}
# the mid-size cases (~20 s each through the oracle pipeline on 8 cores; row tolerance as baseline_cfg0, tests/test_oracle_golden.py)
MID = ("mid_shared_a", "mid_shared_b", "mid_shared_c", "mid_low_overlap", "mid_kitti")


def make_case(bx, name):
    ds, kind, n, seed, identical, patch_ov, match_ov = CASES[name]
    cfg = bx.make_cfg(ds)
    for k, v in patch_ov.items():
        cfg.patch[k] = v
    for k, v in match_ov.items():
        cfg.match[k] = v
    if kind == "indoor":
        pair = bx.synth.make_pair(seed, "indoor", n_target=n, identical=identical)
    elif kind == "shared":
        pair = bx.synth.make_pair(seed, "indoor", n_target=n, shared=True)
    elif kind == "shared_low":
        pair = bx.synth.make_pair(seed, "indoor", n_target=n, shared=True, overlap=0.3)
    elif kind == "outdoor_mid":
        pair = bx.synth.make_pair(seed, "outdoor", voxel=0.15)
    else:
        pair = bx.synth.make_pair(seed, "outdoor", voxel=0.6)
    return cfg, pair, seed


def run_gpu(bx, packed, oracle, cfg, pair, seed):
    from bufferx_amd import lib
    ctx = lib.Context(cfg, max_points=max(len(pair["src"]), len(pair["tgt"])), device=0, packed_weights=packed)
    S = cfg.patch.num_scales
    perm_s = np.stack([oracle.make_perm(len(pair["src"]), seed, 2 * i) for i in range(S)])
    perm_t = np.stack([oracle.make_perm(len(pair["tgt"]), seed, 2 * i + 1) for i in range(S)])
    res = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed)
    out = (np.array(res.pose).reshape(4, 4), res.num_inliers, res.num_mutual, res.num_inlier_ind, res.scales_used,
           [res.des_r[i] for i in range(S)])
    ctx.close()
    return out


@pytest.mark.parametrize("name", list(CASES))
def test_pair_matches_oracle_and_golden(bx, packed, oracle, golden_dir, name):
    from oracle import pipeline as PL
    cfg, pair, seed = make_case(bx, name)
    pose, n_inl, n_mut, n_ind, scales, des_r = run_gpu(bx, packed, oracle, cfg, pair, seed)
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed)
    # bit-exact against the oracle (same arithmetic contract)
    assert (n_inl, n_mut, n_ind, scales) == tuple(ref[1:])
    assert np.array_equal(pose, np.asarray(ref[0], np.float64))
    # and within tolerance of the REAL reference code's output (golden fixture)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    assert np.array_equal(pair["src"][:8], g["src_head"]) and np.array_equal(pair["tgt"][:8], g["tgt_head"])
    assert (n_inl, n_ind, scales) == (int(g["num_inliers"]), int(g["num_inlier_ind"]), int(g["scales_used"]))
    # mid-size cases: a match of a keypoint whose descriptor the reference's torch / numpy arithmetic perturbs (ulp-bound patch decision)
    # may differ -- bounded as in tests/test_oracle_golden.py, where the sets are compared member by member
    assert abs(n_mut - int(g["num_mutual"])) <= (3 * scales if name in MID else 0)
    assert np.allclose(des_r[:scales], g["des_r"][:scales], atol=1e-6)
    # north_star tolerance: 1e-4 deg / 1e-4 m
    rre, rte = bx.synth.pose_difference(pose, g["pose"])    # well-conditioned at zero (synth.py)
    assert rre < 1e-4 and rte < 1e-4


@pytest.mark.parametrize("name", ["indoor_early", "indoor_3scale", "indoor_success", "outdoor_small", "baseline_cfg0"])
def test_pair_in_two_calls_equals_one_call(bx, packed, oracle, name):
    """bx_register_pair_begin / _finish (the early-exit decision taken on the host, as the reference takes it: models/BUFFERX.py:424-457)
    against bx_register_pair on the same context: the exit taken at scale 0 (the later scales are never enqueued), armed but not taken,
    switched off (the first call runs every scale), and a one-scale configuration -- every field of the result equal bit for bit, twice
    in a row on the same context (no state of a pair that left survives into the next one), and the protocol errors."""
    from bufferx_amd import lib
    cfg, pair, seed = make_case(bx, name)
    ctx = lib.Context(cfg, max_points=max(len(pair["src"]), len(pair["tgt"])), device=0, packed_weights=packed)
    S = cfg.patch.num_scales
    perm_s = np.stack([oracle.make_perm(len(pair["src"]), seed, 2 * i) for i in range(S)])
    perm_t = np.stack([oracle.make_perm(len(pair["tgt"]), seed, 2 * i + 1) for i in range(S)])

    def fields(r):
        return (np.array(r.pose).tobytes(), r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used, r.ransac_iters, r.status,
                tuple(r.des_r[i] for i in range(S)))
    try:
        with pytest.raises(lib.BxError):
            ctx.register_pair_finish_async(0)          # nothing pending
        one = fields(ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed))
        for _ in range(2):
            two = ctx.register_pair_two_calls(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed)
            assert fields(two) == one
        # the decision the host saw is the one the device took
        import torch
        flag = ctx.register_pair_begin_async(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed, ctx.new_exit_flag())
        torch.cuda.current_stream(0).synchronize()
        took = bool(cfg.match.get("enable_early_exit", False)) and S > 1 and name == "indoor_early"
        assert int(flag[0]) == int(took)
        res = ctx.register_pair_finish_async(int(flag[0]))
        torch.cuda.current_stream(0).synchronize()
        assert fields(res) == one and res.scales_used == (1 if took else S)
        assert fields(ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed)) == one
    finally:
        ctx.close()


def test_pair_larger_config_success(bx, packed, oracle):
    """K=1024, P=256, 2 scales on a 12k-point noisy partial-overlap pair: registration must succeed and agree
    with the oracle bit for bit."""
    from oracle import pipeline as PL
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 1024, 256, 2
    cfg.patch.search_radius_thresholds = [5, 2]
    cfg.patch.num_points_radius_estimate = 512
    pair = bx.synth.make_pair(21, "indoor", n_target=12000, overlap=0.8)
    pose, n_inl, n_mut, n_ind, scales, _ = run_gpu(bx, packed, oracle, cfg, pair, 9)
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], 9)
    assert (n_inl, n_mut, n_ind, scales) == tuple(ref[1:])
    assert np.array_equal(pose, np.asarray(ref[0], np.float64))


def test_pair_kitti_scale_cloud(bx, packed, oracle):
    """BASELINE configs[2] geometry at reduced keypoint count: ~50k-point outdoor LiDAR-like clouds (aligned to the
    global z axis, outdoor match parameters incl. confidence 1.0), multi-workgroup FPS, the 16-words-per-lane
    neighbour bitmap and the large-extent grid; bit-exact against the oracle."""
    from oracle import pipeline as PL
    cfg = bx.make_cfg("KITTI")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 160, 128, 2
    cfg.patch.search_radius_thresholds = [2, 0.5]
    cfg.patch.num_points_radius_estimate = 160
    cfg.match.iter_n = 1500
    pair = bx.synth.make_pair(5, "outdoor", voxel=0.04)
    assert len(pair["src"]) > 40000 and pair["aligned_z"]
    pose, n_inl, n_mut, n_ind, scales, _ = run_gpu(bx, packed, oracle, cfg, pair, 4)
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], 4)
    assert (n_inl, n_mut, n_ind, scales) == tuple(ref[1:])
    assert np.array_equal(pose, np.asarray(ref[0], np.float64))


def test_full_size_properties(bx, packed, oracle):
    """BASELINE configs[1] sizes (K = 5000, P = 1024, 3 scales, N ~ 45k): too large for the oracle pipeline in a test,
    so the full-size run is checked through size-independent properties: determinism (two runs bit-identical),
    FPS indices unique, every gathered neighbour inside the radius and in ascending cloud order, padding rule, and
    spot checks of 40 keypoints' neighbour lists against a brute-force NumPy search."""
    import torch
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 5000, 1024, 3
    cfg.patch.search_radius_thresholds = [5, 2, 0.5]
    pair = bx.synth.make_pair(31, "indoor", n_target=45000)
    K, P = 5000, 1024
    ctx = lib.Context(cfg, max_points=max(len(pair["src"]), len(pair["tgt"])), device=0, packed_weights=packed)
    S = 3
    perm_s = np.stack([oracle.make_perm(len(pair["src"]), 1, 2 * i) for i in range(S)])
    perm_t = np.stack([oracle.make_perm(len(pair["tgt"]), 1, 2 * i + 1) for i in range(S)])
    r1 = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, 1)
    a = (list(r1.pose), r1.num_inliers, r1.num_mutual, r1.num_inlier_ind, r1.scales_used)
    r2 = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, 1)
    assert a == (list(r2.pose), r2.num_inliers, r2.num_mutual, r2.num_inlier_ind, r2.scales_used)
    assert r1.status == 0 and r1.scales_used == 3 and r1.num_mutual > 0
    pts = pair["src"]
    idx, kp = ctx.fps(pts, K)
    idx = idx.cpu().numpy()
    assert len(np.unique(idx)) == K and idx[0] == 0
    assert np.array_equal(kp.cpu().numpy(), pts[idx])
    pp = pts[perm_s[0]]
    rad = float(r1.des_r[0])
    nidx, patches = ctx.ball_group(pp, kp, torch.tensor([rad], dtype=torch.float64), P)
    nidx, patches = nidx.cpu().numpy(), patches.cpu().numpy()
    kpn = kp.cpu().numpy()
    r2f = np.float32(rad) * np.float32(rad)
    for q in np.random.default_rng(0).choice(K, 40, replace=False):
        d = kpn[q] - pp
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        hits = np.nonzero(d2 < r2f)[0][:P]
        exp = np.full(P, hits[0] if len(hits) else 0, np.int32)
        exp[:len(hits)] = hits
        assert np.array_equal(nidx[q], exp)
    first = nidx[:, :1]
    real = np.concatenate([np.ones((K, 1), bool), nidx[:, 1:] != first], 1)
    real[:, P - 1] = False
    assert np.array_equal(patches[real], pp[nidx[real]])
    assert np.array_equal(patches[:, P - 1], kpn)
    # ascending order among the real (non-padded) entries
    inc = np.diff(nidx.astype(np.int64), axis=1)
    assert np.all((inc > 0) | ~real[:, 1:] | ~real[:, :-1] | (np.arange(1, P)[None, :] == P - 1))
    ctx.close()


def test_pair_tiny_clouds(bx, packed, oracle):
    """Clouds smaller than num_fps and than num_points_per_patch (a cropped fragment): the whole path must stay
    well-defined and identical to the oracle (duplicate keypoints, padded patches, few or no mutual matches)."""
    from oracle import pipeline as PL
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 128, 96, 2
    cfg.patch.search_radius_thresholds = [5, 2]
    cfg.patch.num_points_radius_estimate = 128
    cfg.match.iter_n = 500
    pair = bx.synth.make_pair(8, "indoor", n_target=3000)
    rng = np.random.default_rng(0)
    pair["src"] = np.ascontiguousarray(pair["src"][rng.choice(len(pair["src"]), 90, replace=False)])
    pair["tgt"] = np.ascontiguousarray(pair["tgt"][rng.choice(len(pair["tgt"]), 70, replace=False)])
    pose, n_inl, n_mut, n_ind, scales, _ = run_gpu(bx, packed, oracle, cfg, pair, 2)
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], 2)
    assert (n_inl, n_mut, n_ind, scales) == tuple(ref[1:])
    assert np.array_equal(pose, np.asarray(ref[0], np.float64))
