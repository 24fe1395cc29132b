"""GPU parity of the KISS-Matcher pose back-end (reference models/pose_estimator.py:50-82) through the C-ABI:
bx_kiss_solve and bx_register_pair with cfg.match.pose_estimator == "kiss_matcher" against the oracle's restatement
(oracle/bx_oracle.c bxo_kiss_solve), bit for bit -- pose in binary64, core size, rotation / final inlier counts, GNC iterations."""
import numpy as np
import pytest

from test_oracle_math import _kiss_case

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def kctx(bx, packed):
    from bufferx_amd import lib
    cfg = bx.make_cfg("KITTI")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 1200, 64, 3
    cfg.patch.num_points_radius_estimate = 128
    cfg.match.pose_estimator = "kiss_matcher"
    c = lib.Context(cfg, max_points=4096, device=0, packed_weights=packed)
    yield c
    c.close()


@pytest.mark.parametrize("seed,M,frac,noise,structured", [(0, 500, 0.7, 0.02, 0.0), (1, 300, 0.5, 0.02, 0.15), (2, 800, 0.8, 0.02, 0.0),
                                                          (3, 3000, 0.6, 0.05, 0.1), (4, 64, 0.3, 0.0, 0.0), (5, 1500, 0.95, 0.02, 0.0)])
def test_kiss_solve_matches_oracle(kctx, oracle, seed, M, frac, noise, structured):
    import torch
    s, g, Rg, tg, bad = _kiss_case(seed, M, frac, noise, structured)
    rng = np.random.default_rng(seed)
    corr = np.sort(rng.permutation(M)[: max(2, int(0.9 * M))]).astype(np.int32)       # a subset, like inlier_ind
    rT, rinfo = oracle.kiss_solve(s, g, corr, 0.3)
    T, info = kctx.kiss_solve(s, g, corr, torch.tensor([len(corr)], dtype=torch.int32), 3600)
    assert np.array_equal(_np(info), rinfo)
    assert np.array_equal(_np(T).reshape(4, 4), rT)


def test_kiss_solve_gnc_iterations_and_degenerate(kctx, oracle):
    import torch
    s, g, Rg, tg, bad = _kiss_case(5, 400, 0.3, 0.02)
    rng = np.random.default_rng(9)
    loose = np.flatnonzero(~bad)[::3]
    g[loose] += (0.35 * rng.standard_normal((len(loose), 3))).astype(np.float32)
    corr = np.arange(400, dtype=np.int32)
    rT, rinfo = oracle.kiss_solve(s, g, corr, 0.3)
    T, info = kctx.kiss_solve(s, g, corr, torch.tensor([400], dtype=torch.int32), 400)
    assert rinfo[3] > 1 and np.array_equal(_np(info), rinfo) and np.array_equal(_np(T).reshape(4, 4), rT)
    for C in (0, 1, 2):
        rT, rinfo = oracle.kiss_solve(s, g, corr[:C], 0.3)
        T, info = kctx.kiss_solve(s, g, corr, torch.tensor([C], dtype=torch.int32), 400)
        assert np.array_equal(_np(info), rinfo) and np.array_equal(_np(T).reshape(4, 4), rT)
    line = np.stack([np.linspace(0, 5, 50), np.zeros(50), np.zeros(50)], 1).astype(np.float32)
    c50 = np.arange(50, dtype=np.int32)
    rT, rinfo = oracle.kiss_solve(line, line + np.float32(1.0), c50, 0.3)
    T, info = kctx.kiss_solve(line, line + np.float32(1.0), c50, torch.tensor([50], dtype=torch.int32), 50)
    assert np.array_equal(_np(info), rinfo) and np.array_equal(_np(T).reshape(4, 4), rT)


@pytest.mark.parametrize("early", [False, True])
def test_pair_with_kiss_matcher_backend(bx, packed, oracle, early):
    """whole pair with --pose_estimator kiss_matcher (utils/test_args.py:74-80), with and without the early exit, vs the oracle pipeline"""
    from oracle import pipeline as PL
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 256, 128, 2
    cfg.patch.search_radius_thresholds = [2, 1]
    cfg.patch.num_points_radius_estimate = 256
    cfg.match.pose_estimator = "kiss_matcher"
    cfg.match.kiss_resolution = 0.1
    cfg.match.enable_early_exit = early
    cfg.match.early_exit_min_inliers = 5
    pair = bx.synth.make_pair(3, "indoor", n_target=5000, identical=True)
    seed = 3
    ctx = lib.Context(cfg, max_points=len(pair["src"]), device=0, packed_weights=packed)
    perm_s = np.stack([oracle.make_perm(len(pair["src"]), seed, 2 * i) for i in range(2)])
    perm_t = np.stack([oracle.make_perm(len(pair["tgt"]), seed, 2 * i + 1) for i in range(2)])
    res = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed)
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed)
    ctx.close()
    assert (res.num_inliers, res.num_mutual, res.num_inlier_ind, res.scales_used) == tuple(ref[1:])
    assert np.array_equal(np.array(res.pose).reshape(4, 4), np.asarray(ref[0], np.float64))
    rre, rte = bx.synth.pose_error(np.array(res.pose).reshape(4, 4), pair["T_gt"])
    assert rre < 1.0 and rte < 0.05


def test_kiss_requires_context_option(bx, packed):
    import torch
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 64, 32, 1
    cfg.patch.search_radius_thresholds = [2]
    ctx = lib.Context(cfg, max_points=1024, device=0, packed_weights=packed)
    with pytest.raises(lib.BxError):
        ctx.kiss_solve(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), np.arange(4, dtype=np.int32),
                       torch.tensor([4], dtype=torch.int32), 4)
    ctx.close()
