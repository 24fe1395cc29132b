"""GPU parity tests: every stage of the HIP hot path, called through the C-ABI (include/bufferx.h), against the
CPU oracle on the same seeded inputs.  Index / integer outputs must be bit-exact; fp32/fp64 outputs are
bit-exact by the arithmetic contract (see oracle/bx_oracle.c) -- asserted with zero tolerance where the
contract covers the whole stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(bx, K=128, P=96, S=1, nk=128, ds="3DMatch", **kw):
    cfg = bx.make_cfg(ds)
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = K, P, S
    cfg.patch.search_radius_thresholds = [5, 2, 0.5][:S]
    cfg.patch.num_points_radius_estimate = nk
    for k, v in kw.items():
        cfg.match[k] = v
    return cfg


@pytest.fixture(scope="module")
def ctx(bx, packed):
    from bufferx_amd import lib
    c = lib.Context(_cfg(bx, K=256, P=128, S=2, nk=256), max_points=70000, device=0, packed_weights=packed)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_big(bx, packed):
    from bufferx_amd import lib
    c = lib.Context(_cfg(bx, K=256, P=128, S=2, nk=256), max_points=140000, device=0, packed_weights=packed)
    yield c
    c.close()


def _np(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ FPS (row 2/3)
@pytest.mark.parametrize("n,m", [(700, 64), (3000, 256), (9000, 256), (20000, 200), (40000, 160), (66000, 96), (300000, 48)])   # 300k: 19 workgroups, above the former 262 144-point cap
def test_fps_exact(ctx, oracle, n, m):
    rng = np.random.default_rng(n)
    xyz = (rng.random((n, 3), np.float32) * 4 - 1).astype(np.float32)
    idx, kp = ctx.fps(xyz, m)
    ref = oracle.fps(xyz, m)
    assert np.array_equal(_np(idx), ref)
    assert np.array_equal(_np(kp), xyz[ref])


@pytest.mark.parametrize("colocate", ["0", "1"])
@pytest.mark.parametrize("ppt", ["4", "8", "16"])
@pytest.mark.parametrize("prune", ["1", "0"])
def test_fps_protocols_and_tilings(ctx, oracle, monkeypatch, colocate, ppt, prune):
    """Both granule protocols (XCD co-located L2 granules / agent-scope granules in dispatch-order placement), bucket pruning on and off
    (round 5) and every points-per-thread instantiation give the oracle's indices, ties included (BX_FPS_COLOCATE / BX_FPS_PPT /
    BX_FPS_PRUNE are test hooks)."""
    monkeypatch.setenv("BX_FPS_COLOCATE", colocate)
    monkeypatch.setenv("BX_FPS_PPT", ppt)
    monkeypatch.setenv("BX_FPS_PRUNE", prune)
    rng = np.random.default_rng(7)
    xyz = (rng.random((40000, 3), np.float32) * 4 - 1).astype(np.float32)
    xyz[1000:3000] = np.round(xyz[1000:3000] * 2) / 2          # a lattice block: exact distance ties across workgroups
    idx, kp = ctx.fps(xyz, 300)
    ref = oracle.fps(xyz, 300)
    assert np.array_equal(_np(idx), ref)
    assert np.array_equal(_np(kp), xyz[ref])


@pytest.mark.parametrize("k", ["1", "2", "4", "8"])
@pytest.mark.parametrize("case", ["blobs", "duplicates", "lattice", "line", "tiny"])
def test_fps_batched_rounds(ctx, oracle, monkeypatch, k, case):
    """Round 6: a cross-workgroup exchange resolves SEVERAL samples (K candidates per workgroup with their buckets' second keys, a bound
    for everything unlisted; k_fps.hip).  The sample sequence must be the sequential one whatever the candidate count (BX_FPS_K: 1 = one
    sample per exchange, as rounds 1-5) on clouds built to stress the resolution: dense blobs (the largest running distances cluster, a
    sample lowers several other candidates), exact duplicates spread over buckets (keys that differ in the tie-break word only), an
    integer lattice (distance ties everywhere), points on a line (Morton buckets degenerate) and a cloud smaller than one bucket."""
    monkeypatch.setenv("BX_FPS_K", k)
    rng = np.random.default_rng(60 + len(case))
    if case == "blobs":
        c = rng.random((12, 3), np.float32) * 6
        xyz = np.concatenate([c[i] + rng.standard_normal((3000, 3)).astype(np.float32) * 0.05 for i in range(12)] + [rng.random((400, 3), np.float32) * 6])
    elif case == "duplicates":
        base = rng.random((5000, 3), np.float32) * 3
        xyz = np.concatenate([base, base[rng.permutation(5000)[:3000]], base[:2000], rng.random((10000, 3), np.float32) * 3])
    elif case == "lattice":
        g = np.stack(np.meshgrid(np.arange(30), np.arange(30), np.arange(24), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.125
        xyz = g
    elif case == "line":
        t = rng.random(30000).astype(np.float32)
        xyz = np.stack([t * 8, t * 0.5 + 1, np.full_like(t, 2.0)], 1)
    else:
        xyz = rng.random((37, 3), np.float32) + 1
    xyz = np.ascontiguousarray(xyz[rng.permutation(len(xyz))], np.float32)
    m = min(400, len(xyz))
    idx, kp = ctx.fps(xyz, m)
    ref = oracle.fps(xyz, m)
    assert np.array_equal(_np(idx), ref), (case, k, int(np.argmax(_np(idx) != ref)))
    assert np.array_equal(_np(kp), xyz[ref])


def test_fps_beyond_one_xcd(bx, oracle, packed):
    """More than 32 workgroups per cloud cannot share one XCD: dispatch-order placement, agent-scope protocol (600k points)."""
    from bufferx_amd import lib
    rng = np.random.default_rng(600)
    xyz = (rng.random((600000, 3), np.float32) * 50).astype(np.float32)
    c = lib.Context(_cfg(bx, K=64, P=64, S=1, nk=64), max_points=600000, device=0, packed_weights=packed)
    try:
        idx, _ = c.fps(xyz, 40)
        assert np.array_equal(_np(idx), oracle.fps(xyz, 40))
    finally:
        c.close()


def test_fps_ties_and_origin_skip(ctx, oracle):
    # integer lattice => many exact distance ties (tie rule: lower k mod 512, then lower k);
    # points within 1e-3 of the origin are never candidates (upstream `continue`)
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(8), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    g = g * 0.25
    g[5] = [0.01, 0.0, 0.0]
    rng = np.random.default_rng(0)
    g = g[rng.permutation(len(g))]
    idx, _ = ctx.fps(g, 200)
    assert np.array_equal(_np(idx), oracle.fps(g, 200))


# ------------------------------------------------------------------ radius estimation (row 1)
def test_radius(ctx, oracle, bx):
    pair = bx.synth.make_pair(1, "indoor", n_target=6000)
    pts = pair["src"]
    kp = pts[oracle.fps(pts, 256)]
    thr = [5, 2, 0.5]
    got = _np(ctx.radius(pts, len(pts), kp, thr))
    ref = [oracle.radius(pts, len(pts), kp, t) for t in thr]
    assert np.array_equal(got, np.array(ref))


@pytest.mark.parametrize("n", [40000, 51000, 130000])
def test_radius_concentrated_bins(ctx_big, oracle, n):
    """radius_hist_kernel under the worst concentration of its histogram: a cloud that is four point masses, so every (keypoint, point)
    distance falls into a handful of the 8 193 bins (tens of thousands of LDS atomics of a workgroup on one address; distances that sit
    exactly on bin thresholds)"""
    rng = np.random.default_rng(n)
    sites = np.array([[0, 0, 0], [0.8, 0.1, 0], [0.2, 1.3, 0.4], [2.0, 2.0, 1.0]], np.float32)
    pts = sites[rng.integers(0, 4, n)]
    pts[:17] += rng.normal(0, 0.05, (17, 3)).astype(np.float32)      # a few points off the masses
    kp = np.ascontiguousarray(pts[rng.permutation(n)[:128]])
    thr = [5, 2, 0.5]
    got = _np(ctx_big.radius(pts, n, kp, thr))
    ref = [oracle.radius(pts, n, kp, t) for t in thr]
    assert np.array_equal(got, np.array(ref))


# ------------------------------------------------------------------ neighbour gather (row 4)
@pytest.mark.parametrize("n,K,P,r", [(5000, 200, 64, 0.35), (5000, 200, 128, 0.2), (30000, 256, 128, 0.12), (2500, 100, 96, 2.0)])
def test_ball_group_exact(ctx, oracle, bx, n, K, P, r):
    import torch
    pair = bx.synth.make_pair(n, "indoor", n_target=n)
    pts = pair["src"]
    perm = oracle.make_perm(len(pts), 5, 0)
    pp = _np(ctx.permute(pts, perm))
    assert np.array_equal(pp, pts[perm])
    kp = pts[oracle.fps(pts, K)]
    rad = torch.tensor([r], dtype=torch.float64)
    idx, patches = ctx.ball_group(pp, kp, rad, P)
    ridx, rpatches = oracle.ball_group(pp, kp, np.float32(r), P)
    assert np.array_equal(_np(idx), ridx)
    assert np.array_equal(_np(patches), rpatches)


@pytest.mark.parametrize("case", ["outdoor", "far_queries", "tiny_radius", "huge_radius", "offset_cloud", "flat", "dup_points", "n63"])
def test_ball_group_grid_edges(ctx, oracle, bx, case):
    """Edge cases of the uniform-grid candidate search: the result must stay the brute-force first-P list."""
    import torch
    import zlib
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 1000)
    P, r = 64, 0.3
    if case == "outdoor":          # large extent / small radius => cell edge grows beyond r (cell budget)
        pts = bx.synth.make_pair(4, "outdoor", n_target=40000)["src"]
        kp = pts[oracle.fps(pts, 128)]
        r = 0.4
    elif case == "far_queries":    # keypoints outside the cloud's bounding box (clamped cell ranges)
        pts = rng.random((4000, 3), np.float32)
        kp = np.concatenate([rng.random((32, 3), np.float32) * 3 - 1, pts[:32] + np.float32(0.29)]).astype(np.float32)
    elif case == "tiny_radius":    # most keypoints only hit themselves
        pts = rng.random((6000, 3), np.float32)
        kp = pts[:100].copy()
        r = 1e-3
    elif case == "huge_radius":    # every point is a hit: first-P truncation, 1-cell grid
        pts = rng.random((3000, 3), np.float32)
        kp = pts[:50].copy()
        r = 10.0
    elif case == "offset_cloud":   # large coordinates: rounding of q -+ r and of the cell map
        pts = (rng.random((8000, 3), np.float32) * 2 + np.float32([1000, -2000, 500])).astype(np.float32)
        kp = pts[::80].copy()
        r = 0.25
    elif case == "flat":           # zero extent along z
        pts = rng.random((5000, 3), np.float32)
        pts[:, 2] = 0.5
        kp = pts[::50].copy()
        r = 0.1
    elif case == "dup_points":     # identical points (same cell, same distance)
        base = rng.random((500, 3), np.float32)
        pts = np.repeat(base, 8, axis=0)[rng.permutation(4000)]
        kp = base[:64].copy()
        r = 0.15
    else:                          # n not a multiple of 64 and smaller than a wave
        pts = rng.random((63, 3), np.float32)
        kp = pts[:10].copy()
        r = 0.5
    pts = np.ascontiguousarray(pts, np.float32)
    idx, patches = ctx.ball_group(pts, kp, torch.tensor([r], dtype=torch.float64), P)
    ridx, rp = oracle.ball_group(pts, kp, np.float32(r), P)
    assert np.array_equal(_np(idx), ridx)
    assert np.array_equal(_np(patches), rp)


def test_ball_group_boundary_distances(ctx, oracle):
    """Points at distance r(1 +- few ulp) from the keypoint along each axis, with the keypoint next to a cell face:
    the strict fp32 test decides, never the grid."""
    import torch
    rng = np.random.default_rng(9)
    r = np.float32(0.25)
    kp = (rng.random((64, 3), np.float32) * 2).astype(np.float32)
    pts = [rng.random((2000, 3), np.float32) * 2]
    for ax in range(3):
        for sgn in (-1, 1):
            for ulps in (-3, -1, 0, 1, 3):
                d = r
                for _ in range(abs(ulps)):
                    d = np.nextafter(d, np.float32(1e9 if ulps > 0 else 0), dtype=np.float32)
                q = kp.copy()
                q[:, ax] = q[:, ax] + np.float32(sgn) * d
                pts.append(q)
    pts = np.concatenate(pts).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    idx, patches = ctx.ball_group(pts, kp, torch.tensor([float(r)], dtype=torch.float64), 128)
    ridx, rp = oracle.ball_group(pts, kp, r, 128)
    assert np.array_equal(_np(idx), ridx) and np.array_equal(_np(patches), rp)


def test_ball_group_no_hits(ctx, oracle):
    import torch
    rng = np.random.default_rng(1)
    pts = rng.random((1000, 3), np.float32)
    kp = (rng.random((16, 3), np.float32) + 10).astype(np.float32)   # far away: no hit -> idx 0 everywhere
    idx, patches = ctx.ball_group(pts, kp, torch.tensor([0.1], dtype=torch.float64), 32)
    ridx, rp = oracle.ball_group(pts, kp, np.float32(0.1), 32)
    assert np.array_equal(_np(idx), ridx) and np.array_equal(_np(patches), rp)


# ------------------------------------------------------------------ patch features (rows 5-8)
@pytest.mark.parametrize("aligned", [False, True])
def test_patch_features(ctx, oracle, bx, packed, aligned):
    import torch
    from bufferx_amd import lib
    pair = bx.synth.make_pair(2, "indoor", n_target=5000)
    pts = pair["src"]
    kp = pts[oracle.fps(pts, 96)]
    r = 0.3
    _, patches = oracle.ball_group(pts, kp, np.float32(r), 128)
    R, feat = ctx.patch_features(patches, torch.tensor([r], dtype=torch.float64), aligned)
    rR, rfeat = oracle.patch_features(patches, r, aligned, packed["pnt_w"], packed["pnt_b"])
    assert np.array_equal(_np(R), rR)
    got = lib.chunked_to_logical(_np(feat))
    assert np.array_equal(got, rfeat)


# ------------------------------------------------------------------ convolution layers (rows 9, 12)
@pytest.mark.parametrize("layer", range(8))
def test_desc_conv_layer_exact(ctx, oracle, bx, packed, layer):
    from bufferx_amd import lib
    L = packed["desc"][layer]
    nch = L["W"].shape[0]
    rng = np.random.default_rng(layer)
    units = 11   # not a multiple of the units-per-workgroup: exercises the tail
    x = rng.standard_normal((units, nch, 140, 16)).astype(np.float32)
    x[rng.random(x.shape) < 0.3] = 0
    ref = oracle.desc_conv(x, bx.weights.cyl_tap_table(), L["W"], L["b"], L["relu"])
    out = ctx.conv_layer(0, layer, lib.logical_to_chunked(x), ref.shape)
    assert np.array_equal(lib.chunked_to_logical(_np(out)), ref)


@pytest.mark.parametrize("layer", range(1, 10))
def test_pose_conv_layer_exact(ctx, oracle, bx, packed, layer):
    from bufferx_amd import lib
    L = packed["pose"][layer]
    dims, k, _ = bx.weights.pose_geometry()[layer]
    tap, od = bx.weights.valid_tap_table(dims, k)
    nch = L["W"].shape[0]
    rng = np.random.default_rng(100 + layer)
    units = 37 if layer >= 5 else 5
    x = rng.standard_normal((units, nch, int(np.prod(dims)), 16)).astype(np.float32)
    ref = oracle.pose_conv(layer, x, tap, dims, L["W"], L["b"], L["relu"])      # layers 1..5: the run's pose_conv form
    out = ctx.conv_layer(1, layer, lib.logical_to_chunked(x), ref.shape)
    assert np.array_equal(lib.chunked_to_logical(_np(out)), ref)


@pytest.mark.parametrize("form", ["winograd43", "winograd22", "direct", "winograd43m"])
def test_desc_conv_every_form(oracle, bx, packed, form):
    """bx_params.desc_conv_form: each of the four forms of the Cylindrical_Net layers against ITS restatement, all 8 layers chained
    (13 units: ragged last group of the three-unit F(4x4) kernel and of the two-unit F(2x2) kernel), and the form echoed by the context."""
    from bufferx_amd import lib
    cfg = _cfg(bx, K=64, P=64, S=1, nk=64)
    cfg.arith.desc_conv = form
    c = lib.Context(cfg, max_points=4096, device=0, packed_weights=packed)
    try:
        rng = np.random.default_rng(40)
        x = np.abs(rng.standard_normal((13, 3, 140, 16))).astype(np.float32)
        for layer, L in enumerate(packed["desc"]):
            ref = oracle.desc_conv(x, bx.weights.cyl_tap_table(), L["W"], L["b"], L["relu"], form=form)
            out = c.conv_layer(0, layer, lib.logical_to_chunked(x), ref.shape)
            assert np.array_equal(lib.chunked_to_logical(_np(out)), ref), (form, layer)
            x = ref
        assert c.params.desc_conv_form == bx.config.ARITH_FORMS["desc_conv"].index(form)
    finally:
        c.close()


@pytest.mark.parametrize("units", [1, 2, 3, 4, 7, 16, 17, 33, 100])
def test_desc_conv_mixed_tiles_unit_counts(oracle, bx, packed, units):
    """winograd43m (round 6: F(4x4) tiles on the map rows 0..3, F(3x4) tiles on the rows 4..6): items of 16 column blocks = 3.2 units, so
    every unit count here ends in a different ragged item (1 unit = 5 of 16 blocks ... 100 units = 500 blocks = 31 items + 4 blocks); all
    eight layers on the GPU's own chained input, bit for bit against bxo_conv_wino43m -- and the rows 0..3 bit for bit against the
    all-F(4x4) form's restatement (they are the same tiles)."""
    from bufferx_amd import lib
    cfg = _cfg(bx, K=max(units, 8), P=64, S=1, nk=8)
    cfg.arith.desc_conv = "winograd43m"
    c = lib.Context(cfg, max_points=4096, device=0, packed_weights=packed)
    try:
        rng = np.random.default_rng(400 + units)
        x = np.abs(rng.standard_normal((units, 3, 140, 16))).astype(np.float32)
        for layer, L in enumerate(packed["desc"]):
            ref = oracle.desc_conv(x, bx.weights.cyl_tap_table(), L["W"], L["b"], L["relu"], form="winograd43m")
            out = lib.chunked_to_logical(_np(c.conv_layer(0, layer, lib.logical_to_chunked(x), ref.shape)))
            assert np.array_equal(out, ref), (units, layer, np.argwhere(out != ref)[:4])
            r43 = oracle.desc_conv(x, bx.weights.cyl_tap_table(), L["W"], L["b"], L["relu"], form="winograd43")
            assert np.array_equal(out.reshape(units, -1, 7, 20, 16)[:, :, :4], r43.reshape(units, -1, 7, 20, 16)[:, :, :4])
            x = ref
    finally:
        c.close()


@pytest.mark.parametrize("form", ["winograd43", "winograd22", "direct"])
def test_pose_conv_every_form(oracle, bx, packed, form):
    """bx_params.pose_conv_form: CostNet layers 1..5 in both forms against their restatements."""
    from bufferx_amd import lib
    cfg = _cfg(bx, K=64, P=64, S=1, nk=64)
    cfg.arith.pose_conv = form
    c = lib.Context(cfg, max_points=4096, device=0, packed_weights=packed)
    try:
        for layer in range(1, 6):
            L = packed["pose"][layer]
            dims, k, _ = bx.weights.pose_geometry()[layer]
            tap, od = bx.weights.valid_tap_table(dims, k)
            rng = np.random.default_rng(200 + layer)
            x = rng.standard_normal((19, L["W"].shape[0], int(np.prod(dims)), 16)).astype(np.float32)     # 19 units: ragged last group for G = 2, 3, 8
            ref = oracle.pose_conv(layer, x, tap, dims, L["W"], L["b"], L["relu"], form=form)
            out = c.conv_layer(1, layer, lib.logical_to_chunked(x), ref.shape)
            assert np.array_equal(lib.chunked_to_logical(_np(out)), ref), (form, layer)
    finally:
        c.close()


def test_desc_net(ctx, oracle, bx, packed):
    from bufferx_amd import lib
    rng = np.random.default_rng(3)
    K = 37
    feat = np.abs(rng.standard_normal((K, 3, 140, 16))).astype(np.float32)
    x = feat
    for L in packed["desc"]:
        x = oracle.desc_conv(x, bx.weights.cyl_tap_table(), L["W"], L["b"], L["relu"])
    rdesc, requi = oracle.desc_head(x, packed["pool_w1"], packed["pool_b1"], packed["pool_w2"], packed["pool_b2"])
    desc, equi, xo = ctx.desc_net(lib.logical_to_chunked(feat), want_x=True)
    assert np.array_equal(lib.chunked_to_logical(_np(xo)), x)
    assert np.array_equal(_np(desc), rdesc)
    assert np.array_equal(_np(equi), requi)


# ------------------------------------------------------------------ matching (row 11)
def test_mutual(ctx, oracle):
    rng = np.random.default_rng(4)
    a = rng.standard_normal((250, 32)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a[rng.permutation(250)][:230] + 0.05 * rng.standard_normal((230, 32)).astype(np.float32)
    b[7] = b[3]  # exact duplicate -> tie broken by lowest index
    sm, tm, cnt = ctx.mutual(a, b.astype(np.float32))
    rs, rt, _, _ = oracle.mutual(a, b)
    m = int(_np(cnt)[0])
    assert m == len(rs)
    assert np.array_equal(_np(sm)[:m], rs) and np.array_equal(_np(tm)[:m], rt)


# ------------------------------------------------------------------ CostNet + soft-argmax (row 12)
@pytest.mark.parametrize("form", ["collapsed", "direct"])
def test_pose_net(oracle, bx, packed, form):
    """CostVolume + CostNet + soft-argmax (models/BUFFERX.py:39-69, models/patchnet.py:192-210), logits and ind bit-exact.
    collapsed: layer 0 = bxo_cost_l0 (binary64 P - Q form, k_cost.hip, the default); direct: layer 0 = the fp32 convolution of the
    materialised cost volume (cost_l1_kernel, cfg.arith.cost_l0 = "direct")."""
    import torch
    from bufferx_amd import lib
    rng = np.random.default_rng(5)
    K = 40
    se = rng.standard_normal((K, 140, 32)).astype(np.float32)
    te = rng.standard_normal((K, 140, 32)).astype(np.float32)
    se /= np.linalg.norm(se, axis=2, keepdims=True)
    te /= np.linalg.norm(te, axis=2, keepdims=True)
    te[:8] = se[:8] + 0.01 * rng.standard_normal((8, 140, 32)).astype(np.float32)     # nearly equal maps: P - Q cancels
    m = 23
    sm = rng.permutation(K)[:m].astype(np.int32)
    tm = rng.permutation(K)[:m].astype(np.int32)
    sm[:8] = np.arange(8); tm[:8] = np.arange(8)
    layers = list(zip(packed["pose"], bx.weights.pose_geometry()))
    if form == "collapsed":
        x = oracle.cost_l0(se, te, sm, tm, packed["pose"][0]["W"], packed["pose"][0]["b"])
        layers = layers[1:]
    else:
        x = oracle.cost_volume(se, te, sm, tm)
    first = 10 - len(layers)
    for li, (L, (dims, k, _)) in enumerate(layers):
        tap, _ = bx.weights.valid_tap_table(dims, k)
        x = oracle.pose_conv(first + li, x, tap, dims, L["W"], L["b"], L["relu"])      # the run's pose_conv form (conftest --arith)
    rind = oracle.soft_argmax(x)
    smp = np.zeros(K, np.int32); smp[:m] = sm
    tmp = np.zeros(K, np.int32); tmp[:m] = tm
    cfg = _cfg(bx, K=64, P=64, S=1, nk=64)
    cfg.arith.cost_l0 = form
    c = lib.Context(cfg, max_points=4096, device=0, packed_weights=packed)
    try:
        ind, logits = c.pose_net(se, te, smp, tmp, torch.tensor([m], dtype=torch.int32), K, want_logits=True)
        assert np.array_equal(lib.chunked_to_logical(_np(logits))[:m], x)
        assert np.array_equal(_np(ind)[:m], rind)
    finally:
        c.close()


# ------------------------------------------------------------------ hypotheses + consensus (row 13)
def _random_rot(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)
    return R.astype(np.float32)


def test_hypotheses_and_consensus(ctx, oracle):
    import torch
    rng = np.random.default_rng(6)
    K, m = 200, 150
    sR, tR = _random_rot(rng, K), _random_rot(rng, K)
    sk = (rng.random((K, 3)) * 3).astype(np.float32)
    tk = (rng.random((K, 3)) * 3).astype(np.float32)
    sm = rng.permutation(K)[:m].astype(np.int32)
    tm = rng.permutation(K)[:m].astype(np.int32)
    ind = (rng.random(m) * 19).astype(np.float32)
    ind[0] = 0.0          # exercises kornia's small-angle branch (angle^2 <= 1e-6)
    ind[1] = 1e-4
    rR, rt = oracle.hypotheses(ind, sR[sm], tR[tm], sk[sm], tk[tm])
    smp = np.zeros(K, np.int32); smp[:m] = sm
    tmp = np.zeros(K, np.int32); tmp[:m] = tm
    indp = np.zeros(K, np.float32); indp[:m] = ind
    R, t, ss, tt = ctx.hypotheses(indp, smp, tmp, torch.tensor([m], dtype=torch.int32), K, sR, tR, sk, tk)
    assert np.array_equal(_np(R)[:m], rR) and np.array_equal(_np(t)[:m], rt)
    assert np.array_equal(_np(ss)[:m], sk[sm]) and np.array_equal(_np(tt)[:m], tk[tm])
    # consensus on a planted model: 60% of the matches follow one rigid motion
    Rg = _random_rot(rng, 1)[0].reshape(3, 3)
    tg = np.array([0.3, -0.2, 0.5], np.float32)
    M = 300
    s = (rng.random((M, 3)) * 4 + 1).astype(np.float32)
    g = (s @ Rg.T + tg).astype(np.float32)
    out = rng.random(M) < 0.4
    g[out] += rng.standard_normal((int(out.sum()), 3)).astype(np.float32)
    Rh = np.tile(Rg.reshape(1, 9), (M, 1)).astype(np.float32)
    Rh[out] = _random_rot(rng, int(out.sum()))
    th = (g - np.einsum("mij,mj->mi", Rh.reshape(M, 3, 3), s)).astype(np.float32)
    rinl, rbest, rcnt = oracle.consensus(Rh, th, s, g, 1 / 3)
    inl, cnt, best = ctx.consensus(Rh, th, s, g, torch.tensor([M], dtype=torch.int32), 512)
    c = int(_np(cnt)[0])
    assert c == len(rinl) and int(_np(best)[0]) == rbest
    assert np.array_equal(_np(inl)[:c], rinl)


# ------------------------------------------------------------------ RANSAC + refinement (rows 14, 16)
@pytest.mark.parametrize("conf,iters", [(0.999, 50000), (1.0, 6000)])
def test_ransac_and_refine(bx, packed, oracle, conf, iters):
    import torch
    from bufferx_amd import lib
    cfg = _cfg(bx, K=256, P=64, S=2, nk=64, confidence=conf, iter_n=iters)
    c = lib.Context(cfg, max_points=1000, device=0, packed_weights=packed)
    rng = np.random.default_rng(7)
    M = 400
    Rg = _random_rot(rng, 1)[0].reshape(3, 3).astype(np.float64)
    tg = np.array([0.5, 0.1, -0.4])
    s = (rng.random((M, 3)) * 3).astype(np.float32)
    g = (s @ Rg.T + tg + 0.01 * rng.standard_normal((M, 3))).astype(np.float32)
    bad = rng.random(M) < 0.5
    g[bad] = (rng.random((int(bad.sum()), 3)) * 3).astype(np.float32)
    corr = np.sort(rng.permutation(M)[:300]).astype(np.int32)
    rT, rn, rit = oracle.ransac(s, g, corr, cfg.match.dist_th, cfg.match.similar_th, conf, iters, 1234)
    T, info = c.ransac(s, g, corr, torch.tensor([len(corr)], dtype=torch.int32), 512, 1234)
    info = _np(info)
    assert info[0] == rn and info[1] == rit
    assert np.array_equal(_np(T).reshape(4, 4), rT)
    # refinement from the RANSAC pose
    rTf, rits = oracle.refine(s, g, cfg.match.dist_th, rT.astype(np.float32))
    Tf, its = c.refine(s, g, torch.tensor([M], dtype=torch.int32), 512, rT.astype(np.float32))
    assert int(_np(its)[0]) == rits
    assert np.array_equal(_np(Tf).reshape(4, 4), rTf)
    c.close()


def test_ransac_degenerate(bx, packed, oracle):
    import torch
    from bufferx_amd import lib
    cfg = _cfg(bx, K=64, P=64, S=1, nk=64, iter_n=500)
    c = lib.Context(cfg, max_points=1000, device=0, packed_weights=packed)
    s = np.zeros((10, 3), np.float32)
    T, info = c.ransac(s, s, np.arange(2, dtype=np.int32), torch.tensor([2], dtype=torch.int32), 64, 1)
    assert np.array_equal(_np(T).reshape(4, 4), np.eye(4)) and _np(info)[0] == 0   # C < 3 -> identity, 0 inliers
    c.close()


# ------------------------------------------------------------------ degenerate sizes
def test_fps_more_samples_than_points(ctx, oracle):
    """m > n: once every point has been selected all running distances are 0 and the tie rule decides (upstream keeps
    sampling); the GPU must replay exactly the oracle's sequence."""
    rng = np.random.default_rng(3)
    xyz = (rng.random((150, 3), np.float32) * 2 + 0.5).astype(np.float32)
    idx, kp = ctx.fps(xyz, 256)
    ref = oracle.fps(xyz, 256)
    assert np.array_equal(_np(idx), ref)
    assert np.array_equal(_np(kp), xyz[ref])


def test_ball_group_patch_larger_than_cloud(ctx, oracle):
    """P > n: every keypoint has fewer than P neighbours; padding and the keypoint slot must follow the reference rule."""
    import torch
    rng = np.random.default_rng(4)
    pts = rng.random((90, 3), np.float32)
    kp = pts[:12].copy()
    idx, patches = ctx.ball_group(pts, kp, torch.tensor([0.6], dtype=torch.float64), 128)
    ridx, rp = oracle.ball_group(pts, kp, np.float32(0.6), 128)
    assert np.array_equal(_np(idx), ridx) and np.array_equal(_np(patches), rp)


@pytest.mark.parametrize("n,r,nkp", [(60000, 0.45, 64), (130000, 0.5, 40)])
def test_ball_group_long_candidate_sequences(bx, oracle, packed, n, r, nkp):
    """Dense cloud + large radius: candidate sequences longer than one register block (n = 60k: ~3600 candidates per keypoint,
    chunk table + bit peeling) and longer than the 64-chunk table (n = 130k: ~10k candidates, row walk); 128k+ points also
    exercise the 5-level bitmap."""
    import torch
    from bufferx_amd import lib
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 3), np.float32) * 2).astype(np.float32)
    kp = np.concatenate([pts[rng.choice(n, nkp - 8, replace=False)], (rng.random((8, 3), np.float32) * 2).astype(np.float32)])
    c = lib.Context(_cfg(bx, K=256, P=128, S=2, nk=256), max_points=n, device=0, packed_weights=packed)
    try:
        idx, patches = c.ball_group(pts, kp, torch.tensor([r], dtype=torch.float64), 128)
        ridx, rp = oracle.ball_group(pts, kp, np.float32(r), 128)
        assert np.array_equal(_np(idx), ridx) and np.array_equal(_np(patches), rp)
    finally:
        c.close()


def test_fps_kitti_size_cloud(bx, oracle, packed):
    """BASELINE configs[2]: ~120k-point clouds -> 8 cooperating FPS workgroups per cloud (granule exchange), bit-exact indices."""
    from bufferx_amd import lib
    rng = np.random.default_rng(120)
    xyz = (rng.normal(size=(125000, 3)) * [30, 30, 2]).astype(np.float32)
    c = lib.Context(_cfg(bx, K=256, P=128, S=2, nk=256), max_points=130000, device=0, packed_weights=packed)
    try:
        idx, kp = c.fps(xyz, 96)
        ref = oracle.fps(xyz, 96)
        assert np.array_equal(_np(idx), ref) and np.array_equal(_np(kp), xyz[ref])
    finally:
        c.close()
