"""GPU: the latency form of bx_register_pair (bx_params.keypoint_tiles > 1) -- furthest point sampling cut into tiles on the
context's own stream, the descriptors of a tile computed beside the sampling of the next -- returns the throughput form's result
bit for bit, and the resumable FPS launches reproduce the single launch (oracle = bxo_fps, pointnet2 semantics)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(bx, K, P, S, thr, nk, early):
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = K, P, S
    cfg.patch.search_radius_thresholds = thr
    cfg.patch.num_points_radius_estimate = nk
    cfg.match.iter_n = 2000
    cfg.match.enable_early_exit = early
    return cfg


def _fields(r):
    return (tuple(r.pose), r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used, r.ransac_iters, r.refine_iters,
            tuple(r.des_r))


@pytest.mark.parametrize("mode", ["off", "taken", "not_taken"])
@pytest.mark.parametrize("tiles", [2, 3, 5])
def test_tiled_pair_equals_untiled(bx, packed, tiles, mode):
    from bufferx_amd import lib
    K, P, S, nk = 400, 128, 3, 96
    cfg = _cfg(bx, K, P, S, [5, 2, 0.5], nk, mode != "off")
    cfg.match.early_exit_min_inliers = 10 ** 6 if mode == "not_taken" else 0
    pair = bx.synth.make_pair(11, "indoor", n_target=40000, shared=True)      # > 32768 points: FPS over 3 workgroups per cloud
    ns, nt = len(pair["src"]), len(pair["tgt"])
    rng = np.random.default_rng(5)
    ps = np.stack([rng.permutation(ns) for _ in range(S)]).astype(np.int32)
    pt = np.stack([rng.permutation(nt) for _ in range(S)]).astype(np.int32)
    out = []
    for t in (0, tiles):
        c = copy.deepcopy(cfg)
        c.test.keypoint_tiles = t
        ctx = lib.Context(c, max_points=max(ns, nt), device=0, packed_weights=packed)
        for _ in range(2):                      # twice on the same context: state carried between pairs must not leak
            r = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], ps, pt, 77)
            out.append(_fields(r))
        ctx.close()
    assert out[0] == out[1] == out[2] == out[3]
    assert out[0][2] > 0
    assert out[0][4] == (1 if mode == "taken" else S)


@pytest.mark.parametrize("early", [False, True])
def test_streams_without_tiles(bx, packed, early):
    """num_fps <= num_points_radius_estimate (the reference's default 1500 < 2000) leaves nothing to tile: the latency form still
    runs the source / target / matching chains on the context's streams and must return the same result."""
    from bufferx_amd import lib
    K, P, S, nk = 200, 96, 2, 256
    cfg = _cfg(bx, K, P, S, [5, 1], nk, early)
    cfg.match.early_exit_min_inliers = 10 ** 6
    pair = bx.synth.make_pair(6, "indoor", n_target=9000, shared=True)
    ns, nt = len(pair["src"]), len(pair["tgt"])
    rng = np.random.default_rng(3)
    ps = np.stack([rng.permutation(ns) for _ in range(S)]).astype(np.int32)
    pt = np.stack([rng.permutation(nt) for _ in range(S)]).astype(np.int32)
    out = []
    for t in (0, 2):
        c = copy.deepcopy(cfg)
        c.test.keypoint_tiles = t
        ctx = lib.Context(c, max_points=max(ns, nt), device=0, packed_weights=packed)
        out.append(_fields(ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], ps, pt, 12)))
        ctx.close()
    assert out[0] == out[1] and out[0][2] > 0 and out[0][4] == S


def test_tiled_pair_equals_oracle(bx, packed, oracle):
    """The latency form against the CPU oracle pipeline directly (not only against the other GPU form)."""
    from bufferx_amd import lib
    from oracle import pipeline as PL
    K, P, S, nk = 192, 96, 2, 64
    cfg = _cfg(bx, K, P, S, [2, 1], nk, False)
    cfg.test.keypoint_tiles = 3
    pair = bx.synth.make_pair(4, "indoor", n_target=5000, shared=True)
    ns, nt = len(pair["src"]), len(pair["tgt"])
    rng = np.random.default_rng(9)
    ps = np.stack([rng.permutation(ns) for _ in range(S)]).astype(np.int32)
    pt = np.stack([rng.permutation(nt) for _ in range(S)]).astype(np.int32)
    ctx = lib.Context(cfg, max_points=max(ns, nt), device=0, packed_weights=packed)
    r = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], ps, pt, 31)
    ctx.close()
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], 31, perms=(ps, pt))
    assert (r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used) == tuple(ref[1:])
    assert np.array_equal(np.array(r.pose, np.float64).reshape(4, 4), np.asarray(ref[0], np.float64))


def test_capture_refused_in_latency_form(bx, packed):
    from bufferx_amd import lib
    cfg = _cfg(bx, 128, 64, 1, [2], 32, False)
    cfg.test.keypoint_tiles = 2
    pair = bx.synth.make_pair(2, "indoor", n_target=3000, shared=True)
    ns, nt = len(pair["src"]), len(pair["tgt"])
    ctx = lib.Context(cfg, max_points=max(ns, nt), device=0, packed_weights=packed)
    with pytest.raises(lib.BxError):       # refused when the capture is set, not later inside bx_register_pair
        ctx.set_capture(0, 0, max(ns, nt))
    ctx.close()
