"""Degenerate correspondence counts through bx_register_pair, as named tests (round 4).

Where the reference raises or hands an ill-posed problem to a library, the product has a documented behaviour (LABBOOK.md section 4):
 * m = 1 mutual match (reference: CostVolume squeezes a batch of one away, models/BUFFERX.py:66): CostNet runs on the one match, the pair
   continues;
 * C = |inlier_ind| < 3 (reference: Open3D's registration_ransac_based_on_correspondence with fewer than ransac_n = 3 correspondences
   returns its default result, models/pose_estimator.py:84-117): identity pose, 0 RANSAC inliers, refinement from the identity;
 * m = 0 cannot be produced by mutual matching of two non-empty descriptor sets (the globally closest pair is always mutual), so it is
   exercised at the stage entry points the whole-pair call is made of: zero matches through CostNet / hypotheses / consensus / RANSAC
   leave the outputs untouched and the counts at zero (reference: torch.argmax of an empty tensor raises, models/BUFFERX.py:415).
Every case is compared with the oracle pipeline bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _cfg(bx, K, P=64, S=2, nk=None, **match):
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = K, P, S
    cfg.patch.search_radius_thresholds = [5, 2, 0.5][:S]
    cfg.patch.num_points_radius_estimate = nk or 64
    cfg.match.iter_n = 500
    for k, v in match.items():
        cfg.match[k] = v
    return cfg


def _run(bx, packed, oracle, cfg, pair, seed):
    from bufferx_amd import lib
    from oracle import pipeline as PL
    ctx = lib.Context(cfg, max_points=max(len(pair["src"]), len(pair["tgt"])), device=0, packed_weights=packed)
    S = cfg.patch.num_scales
    ps = np.stack([oracle.make_perm(len(pair["src"]), seed, 2 * i) for i in range(S)])
    pt = np.stack([oracle.make_perm(len(pair["tgt"]), seed, 2 * i + 1) for i in range(S)])
    r = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], ps, pt, seed)
    got = (np.array(r.pose).reshape(4, 4), r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used, r.status)
    ctx.close()
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed)
    return got, ref


def test_single_mutual_match(bx, packed, oracle):
    """num_fps = 1: one keypoint per cloud -> exactly one mutual match per scale (m = 1), M = 2 over two scales, C <= 2 < 3: identity
    pose, no RANSAC inliers; the one-match CostNet / hypothesis / consensus path equals the oracle's."""
    cfg = _cfg(bx, K=1, nk=64)
    pair = bx.synth.make_pair(4, "indoor", n_target=3000, shared=True)
    got, ref = _run(bx, packed, oracle, cfg, pair, 3)
    assert got[5] == 0 and got[2] == 2 and got[4] == 2            # one match per scale
    assert got[1:5] == tuple(ref[1:])
    assert np.array_equal(got[0], np.asarray(ref[0], np.float64))
    assert got[3] < 3 and got[1] == 0                             # C < 3 -> Open3D's default result


@pytest.mark.parametrize("K", [2, 3, 5])
def test_fewer_than_three_consensus_members(bx, packed, oracle, K):
    """A handful of keypoints: the consensus set cannot reach three members on unrelated fragments -> _estimate_ransac's degenerate
    branch (identity, 0 inliers), then post_refinement from the identity -- the same as the oracle."""
    cfg = _cfg(bx, K=K, nk=64)
    pair = bx.synth.make_pair(6, "indoor", n_target=3000, overlap=0.2)       # independently sampled low-overlap fragments
    got, ref = _run(bx, packed, oracle, cfg, pair, 5)
    assert got[5] == 0 and got[1:5] == tuple(ref[1:])
    assert np.array_equal(got[0], np.asarray(ref[0], np.float64))
    if got[3] < 3:
        assert got[1] == 0


def test_low_overlap_tiny_consensus(bx, packed, oracle):
    """3DLoMatch-like pair (15 % overlap) at a size the oracle runs: a consensus set of a few members, RANSAC on it -- bit-identical."""
    cfg = _cfg(bx, K=192, P=96, S=2, nk=128)
    cfg.match.iter_n = 4000
    pair = bx.synth.make_pair(10, "indoor", n_target=5000, shared=True, overlap=0.2)
    got, ref = _run(bx, packed, oracle, cfg, pair, 8)
    assert got[5] == 0 and got[1:5] == tuple(ref[1:])
    assert np.array_equal(got[0], np.asarray(ref[0], np.float64))
    assert got[1:5] == (4, 132, 4, 2)        # four consensus members, all four RANSAC inliers (what the oracle finds on this pair)


def test_zero_matches_through_the_stages(bx, packed, oracle):
    """m = 0 / M = 0 / C = 0 at the stage entry points: nothing is written, every count stays zero, no kernel faults."""
    import torch
    from bufferx_amd import lib
    cfg = _cfg(bx, K=64, nk=64)
    c = lib.Context(cfg, max_points=4096, device=0, packed_weights=packed)
    try:
        rng = np.random.default_rng(0)
        K = 64
        se = rng.standard_normal((K, 140, 32)).astype(np.float32)
        te = rng.standard_normal((K, 140, 32)).astype(np.float32)
        zero = torch.zeros(1, dtype=torch.int32)
        mids = np.zeros(K, np.int32)
        ind, logits = c.pose_net(se, te, mids, mids, zero, K, want_logits=True)        # zero matches: no unit runs
        torch.cuda.synchronize()
        R = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (K, 1))
        z3 = np.zeros((K, 3), np.float32)
        inl, cnt, best = c.consensus(R, z3, z3, z3, zero, K)
        assert int(_np(cnt)[0]) == 0
        T, info = c.ransac(z3, z3, np.zeros(K, np.int32), zero, K, 1)
        assert np.array_equal(_np(T).reshape(4, 4), np.eye(4)) and int(_np(info)[0]) == 0
        for C_ in (1, 2):
            T, info = c.ransac(z3, z3, np.arange(K, dtype=np.int32), torch.tensor([C_], dtype=torch.int32), K, 1)
            assert np.array_equal(_np(T).reshape(4, 4), np.eye(4)) and int(_np(info)[0]) == 0
    finally:
        c.close()
