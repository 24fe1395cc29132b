"""GPU parity at the HEADLINE configuration (BASELINE configs[1]: K = 5000 keypoints, P = 1024 points per patch, 3 scales) and
at configs[0] (1 scale, 512 keypoints, 512 points per patch).

The CPU oracle cannot run a whole K = 5000 pair inside a test, so bx_register_pair runs with bx_set_capture armed and every
stage is checked on a random sample of its units against the oracle stage FED THE GPU'S OWN UPSTREAM TENSORS -- a stage whose
output differed from the oracle's on the same input would be caught wherever in the chain it sits.  Cheap stages (mutual
matching at 5000 x 5000, hypotheses, consensus at M ~ 4000..15000, RANSAC, refinement) are checked in full.  All comparisons are
bit-exact (the arithmetic contract of oracle/bx_oracle.c).  Reference: models/BUFFERX.py:257-467, models/patch_embedder.py:44-170,
models/patchnet.py:49-84,184-210, models/pose_estimator.py:84-117."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, P, S = 5000, 1024, 3
NSAMP = 64

# The three real-size configurations, on exactly the inputs of the reference-minted fixtures tests/golden/<name>.npz
# (tests/golden/make_golden.py ran the reference's own BufferX.forward on them at this size):
#   headline_cfg1  BASELINE configs[1]: 3DMatch configuration, ~25k / 30k-point partial-overlap fragments, RANSAC (conf 0.999) + refinement
#   kitti_cfg2     BASELINE configs[2]: KITTI configuration (config/outdoor_config.py:57-70: is_aligned_to_global_z -> R = I branch of
#                  models/patch_embedder.py:142-146, confidence 1.0 -> all 50 000 RANSAC iterations, no refinement -> binary64 pose),
#                  two ~90k-point LiDAR sweeps
#   tiers_early    BASELINE configs[4]: TIERS_hetero configuration (config/tiers_hetero_config.py:9 = outdoor parameters) with
#                  enable_early_exit (models/BUFFERX.py:424-439), 107k-point dense sweep vs its 54k-point sparse subset; exit TAKEN
#   headline_cfg1_b / _c, kitti_cfg2_b (round 4): two more of bench.py's 3DMatch-like pairs (seeds 103 and 112; ~37k / 32k and ~27k / 28k
#                  points) and a second pair of LiDAR sweeps
BIG = {
    "headline_cfg1": ("3DMatch", dict(), dict(), ("shared", 100, 30000)),
    "kitti_cfg2": ("KITTI", dict(), dict(), ("kitti", 100, 0)),
    "tiers_early": ("TIERS_hetero", dict(), dict(enable_early_exit=True), ("tiers", 100, 0)),
    "headline_cfg1_b": ("3DMatch", dict(), dict(), ("shared", 103, 36885)),
    "headline_cfg1_c": ("3DMatch", dict(), dict(), ("shared", 112, 28334)),
    "kitti_cfg2_b": ("KITTI", dict(), dict(), ("kitti", 101, 0)),
    # a 3DLoMatch-like pair at the real size (20 % overlap, the reference's 3DLoMatch configuration): it does NOT register -- the consensus
    # set sits on the 180-degree twin of the near-symmetric room -- and the reference's (wrong) answer has to be reproduced all the same
    "headline_lo": ("3DLoMatch", dict(), dict(), ("shared_lo20", 120, 35000)),
}
# Matches that differ from the reference's run, PINNED per fixture at what was observed (profiles/r04_realsize_report.jsonl, re-measured
# every round): a regression that flips more matches than this fails, on the fixtures that had none as much as on the two that have some.
PINNED_FLIPS = {"headline_cfg1": 0, "kitti_cfg2": 0, "tiers_early": 0, "headline_cfg1_b": 4, "headline_cfg1_c": 0, "kitti_cfg2_b": 0,
                "headline_lo": 1}
assert set(PINNED_FLIPS) == set(BIG)


def golden_path(name):
    """A missing fixture is a FAILURE, not a silently smaller suite (the seven names above are part of the parity claim)."""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    assert os.path.exists(p), f"real-size fixture {p} is missing: re-mint it with tests/golden/make_golden.py {name}"
    return p


def _np(t):
    return t.detach().cpu().numpy()


def _headline_cfg(bx):
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = K, P, S
    cfg.patch.search_radius_thresholds = [5, 2, 0.5]
    return cfg


def big_case(bx, name):
    """(cfg, pair, seed) of a real-size case == tests/golden/make_golden.py::case_inputs(name) + its overrides."""
    ds, patch_ov, match_ov, (kind, seed, n) = BIG[name]
    cfg = bx.make_cfg(ds)
    cfg.patch.num_fps, cfg.patch.num_points_per_patch = K, P
    assert cfg.patch.num_scales == S and list(cfg.patch.search_radius_thresholds) == [5, 2, 0.5]
    for k, v in patch_ov.items():
        cfg.patch[k] = v
    for k, v in match_ov.items():
        cfg.match[k] = v
    if kind == "shared":
        pair = bx.synth.make_pair(seed, "indoor", n_target=n, shared=True)
    elif kind == "shared_lo20":
        pair = bx.synth.make_pair(seed, "indoor", n_target=n, shared=True, overlap=0.2)
    elif kind == "kitti":
        pair = bx.synth.make_pair(seed, "outdoor", voxel=0.02)
    else:
        pair = bx.synth.make_tiers_pair(seed)
    return cfg, pair, seed


@pytest.fixture(scope="module", params=list(BIG))
def headline(request, bx, packed, oracle):
    """One real-size pair (K = 5000 / P = 1024 / S = 3) run once per scale with the capture armed on (scale 0, src), (scale 1, tgt),
    (scale 2, src) and once without: the result must not depend on the capture."""
    import torch
    from bufferx_amd import lib
    name = request.param
    golden_path(name)
    cfg, pair, seed = big_case(bx, name)
    ctx = lib.Context(cfg, max_points=max(len(pair["src"]), len(pair["tgt"])), device=0, packed_weights=packed)
    perm = [np.stack([oracle.make_perm(len(pair[k]), seed, 2 * i + c).astype(np.int32) for i in range(S)]) for c, k in enumerate(("src", "tgt"))]
    runs = []
    for scale, cloud in ((0, 0), (1, 1), (2, 0)):
        n = len(pair["src" if cloud == 0 else "tgt"])
        cap = ctx.set_capture(scale, cloud, n)
        res = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm[0], perm[1], seed)
        torch.cuda.synchronize()
        out = dict(scale=scale, cloud=cloud, cap=cap, pose=np.array(res.pose).reshape(4, 4),
                   tup=(res.num_inliers, res.num_mutual, res.num_inlier_ind, res.scales_used, res.ransac_iters, res.refine_iters),
                   des_r=[float(res.des_r[i]) for i in range(S)], status=res.status)
        runs.append(out)
    ctx.set_capture(None, 0, 0)
    plain = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm[0], perm[1], seed)
    plain = (np.array(plain.pose).reshape(4, 4), (plain.num_inliers, plain.num_mutual, plain.num_inlier_ind, plain.scales_used,
                                                   plain.ransac_iters, plain.refine_iters))
    used = plain[1][3]
    yield dict(name=name, cfg=cfg, pair=pair, perm=perm, seed=seed, runs=runs, plain=plain, ctx=ctx, used=used)
    ctx.close()


def test_headline_runs_agree(headline):
    """capture on / off and the three captured runs give the identical result; the expected scales ran (all three, or one when the
    early exit of the TIERS case is taken)."""
    p0, t0 = headline["plain"]
    for r in headline["runs"]:
        assert r["status"] == 0
        assert np.array_equal(r["pose"], p0) and r["tup"] == t0
    assert t0[3] == (1 if headline["name"] == "tiers_early" else S) and t0[1] > 0
    if headline["name"].startswith("kitti_cfg2"):
        assert t0[4] == headline["cfg"].match.iter_n       # confidence 1.0: every one of the 50 000 iterations is visited


@pytest.mark.parametrize("ri", [0, 1, 2])
def test_headline_descriptor_chain(headline, oracle, packed, bx, ri):
    """permute -> select_patches -> axis_align / SPT / pnt_layer -> Cylindrical_Net (8 layers) -> pool_layer + norms on NSAMP random
    keypoints of the captured (scale, cloud), each stage fed the GPU tensor of the stage before it."""
    import torch
    from bufferx_amd import lib
    W = bx.weights
    run, cfg, pair = headline["runs"][ri], headline["cfg"], headline["pair"]
    cap, scale, cloud = run["cap"], run["scale"], run["cloud"]
    if scale >= headline["used"]:
        pytest.skip("early exit taken: this scale never ran")
    cloud_pts = pair["src" if cloud == 0 else "tgt"]
    pts_perm = _np(cap["pts_perm"])
    assert np.array_equal(pts_perm, cloud_pts[headline["perm"][cloud][scale]])
    kpts = _np(cap["kpts"][cloud])
    assert np.array_equal(kpts, cloud_pts[oracle.fps(cloud_pts, K)])                       # FPS at K = 5000, in full
    des_r = run["des_r"][scale]
    rng = np.random.default_rng(100 + ri)
    sel = np.sort(rng.choice(K, NSAMP, replace=False))
    tsel = torch.as_tensor(sel, device=cap["patches"].device)
    # select_patches
    g_patches = _np(cap["patches"][tsel])
    _, o_patches = oracle.ball_group(pts_perm, kpts[sel], np.float32(des_r), P)
    assert np.array_equal(g_patches, o_patches)
    # patch features
    o_R, o_feat = oracle.patch_features(g_patches, des_r, pair["aligned_z"], packed["pnt_w"], packed["pnt_b"])
    assert np.array_equal(_np(cap["R"][cloud][tsel]), o_R.reshape(-1, 9))
    g_feat = lib.chunked_to_logical(_np(cap["feat"][tsel]))         # the oracle keeps logical channel order
    assert np.array_equal(g_feat, o_feat)
    # conv stack (the per-layer check at persistent-walk sizes is test_conv_layers_group_walk)
    tap = W.cyl_tap_table(cfg.patch.ele_n, cfg.patch.azi_n)
    x = g_feat
    for L in packed["desc"]:
        x = oracle.desc_conv(x, tap, L["W"], L["b"], L["relu"])
    g_x = lib.chunked_to_logical(_np(cap["x"][tsel]))
    assert np.array_equal(g_x, x)
    # head
    o_desc, o_equi = oracle.desc_head(g_x, packed["pool_w1"], packed["pool_b1"], packed["pool_w2"], packed["pool_b2"])
    assert np.array_equal(_np(cap["desc"][cloud][tsel]), o_desc)
    assert np.array_equal(_np(cap["equi"][cloud][tsel]), o_equi)


@pytest.mark.parametrize("ri", [0, 1, 2])
def test_headline_matching_chain(headline, oracle, packed, bx, ri):
    """mutual matching over all 5000 x 5000 descriptors, CostVolume + CostNet + soft-argmax on NSAMP random matches, hypotheses and
    the cumulative consensus (every hypothesis, every correspondence) of the captured scale."""
    import torch
    from oracle import pipeline as PL
    run, cfg = headline["runs"][ri], headline["cfg"]
    cap, scale = run["cap"], run["scale"]
    if scale >= headline["used"]:
        pytest.skip("early exit taken: this scale never ran")
    m, M, C, best = [int(v) for v in _np(cap["counts"])]
    d0, d1 = _np(cap["desc"][0]), _np(cap["desc"][1])
    o_s, o_t, _, _ = oracle.mutual(d0, d1)
    assert m == len(o_s) and m > 64
    s_mids, t_mids = _np(cap["s_mids"])[:m], _np(cap["t_mids"])[:m]
    assert np.array_equal(s_mids, o_s) and np.array_equal(t_mids, o_t)
    # CostNet on a sample of the matches: gather the two equivariant maps of the sampled matches on the device
    rng = np.random.default_rng(200 + ri)
    sel = np.sort(rng.choice(m, NSAMP, replace=False))
    dev = cap["equi"][0].device
    e0 = _np(cap["equi"][0][torch.as_tensor(s_mids[sel].astype(np.int64), device=dev)])
    e1 = _np(cap["equi"][1][torch.as_tensor(t_mids[sel].astype(np.int64), device=dev)])
    ar = np.arange(NSAMP, dtype=np.int32)
    o_ind = PL.pose_forward(e0, e1, ar, ar, packed, cfg)
    g_ind = _np(cap["ind"])[:m]
    assert np.array_equal(g_ind[sel], o_ind)
    # hypotheses of this scale = the last m rows of the accumulated tensors
    R0, R1 = _np(cap["R"][0]), _np(cap["R"][1])
    k0, k1 = _np(cap["kpts"][0]), _np(cap["kpts"][1])
    o_R, o_tr = oracle.hypotheses(g_ind, R0[s_mids], R1[t_mids], k0[s_mids], k1[t_mids], cfg.patch.azi_n)
    R_cat, t_cat = _np(cap["R_cat"])[:M], _np(cap["t_cat"])[:M]
    ss, tt = _np(cap["ss_cat"])[:M], _np(cap["tt_cat"])[:M]
    assert np.array_equal(R_cat[M - m:], o_R) and np.array_equal(t_cat[M - m:], o_tr)
    assert np.array_equal(ss[M - m:], k0[s_mids]) and np.array_equal(tt[M - m:], k1[t_mids])
    # consensus over everything accumulated so far
    o_inl, o_best, o_counts = oracle.consensus(R_cat, t_cat, ss, tt, cfg.match.inlier_th, cfg.patch.azi_n)
    assert best == o_best and C == len(o_inl)
    assert np.array_equal(_np(cap["cons_cnt"])[:M], o_counts)
    assert np.array_equal(_np(cap["inlier_ind"])[:C], o_inl)
    if scale == headline["used"] - 1:
        assert M == run["tup"][1] and C == run["tup"][2]


def test_headline_pose(headline, oracle):
    """RANSAC (fp64, seeded; all 50 000 iterations in the outdoor configurations) and -- where the configuration refines --
    post_refinement from the captured correspondences of the last scale that ran == the returned pose."""
    used = headline["used"]
    run, cfg, seed = headline["runs"][used - 1], headline["cfg"], headline["seed"]
    cap = run["cap"]
    m, M, C, best = [int(v) for v in _np(cap["counts"])]
    ss, tt = _np(cap["ss_cat"])[:M], _np(cap["tt_cat"])[:M]
    inl = _np(cap["inlier_ind"])[:C]
    # one pose-estimation call in every case: without the early exit it is the final one; with the exit TAKEN it is the test call
    T, n, it = oracle.ransac(ss, tt, inl, cfg.match.dist_th, cfg.match.similar_th, cfg.match.confidence, cfg.match.iter_n,
                             oracle.mix64(seed, 0x5AC0000))
    assert np.array_equal(_np(cap["T_ransac"]).reshape(4, 4), T)
    assert (n, it) == (run["tup"][0], run["tup"][4])
    if cfg.test.pose_refine is True:
        Tr, rit = oracle.refine(ss, tt, cfg.match.dist_th, T.astype(np.float32))
        assert np.array_equal(run["pose"], Tr.reshape(4, 4).astype(np.float64)) and rit == run["tup"][5]
    else:
        assert np.array_equal(run["pose"], T)          # un-refined binary64 pose (every outdoor configuration)
    if headline["name"] == "tiers_early":
        assert n >= cfg.match.early_exit_min_inliers   # the exit was taken because of this count


def test_headline_pair_registers(headline, bx):
    """the pair is actually registered (indoor: RTE < 0.3 m, RRE < 15 deg, config/indoor_config.py:36-37; outdoor: 2 m / 5 deg,
    config/outdoor_config.py:36-37)"""
    cfg = headline["cfg"]
    rre, rte = bx.synth.pose_error(headline["plain"][0], headline["pair"]["T_gt"])
    if headline["name"] == "headline_lo":
        # the 20 %-overlap pair is NOT registered, by the reference either (tests/golden/headline_lo.npz: RRE 179.83 deg -- the consensus
        # set sits on the 180-degree twin of the near-symmetric synthetic room); the same wrong answer is the requirement here
        assert rre > 170.0, (rre, rte)
        return
    assert rre < cfg.test.rre_thresh and rte < cfg.test.rte_thresh, (rre, rte, headline["plain"][1])


def test_headline_vs_reference(headline, bx, golden_dir):
    """GPU result against the fixture minted by the REFERENCE's own forward at this size (models/BUFFERX.py:257-467 through
    tests/golden/ref_harness.py): identical counts (RANSAC inliers, accumulated mutual matches, consensus set, scales used), radii,
    per-scale mutual sets and consensus set, and the pose within the north-star tolerance 1e-4 deg / 1e-4 m.  The number of
    descriptor rows that differ beyond 2e-5 (a point within an ulp of a radius / voxel bound decided differently by the reference's
    torch / numpy arithmetic) is reported AND bounded (0.4 % of the sampled rows): LABBOOK.md section 4 quotes it."""
    g = np.load(golden_path(headline["name"]))
    pair, used = headline["pair"], headline["used"]
    assert np.array_equal(pair["src"][:8], g["src_head"]) and np.array_equal(pair["tgt"][:8], g["tgt_head"])
    assert (len(pair["src"]), len(pair["tgt"])) == (int(g["n_src"]), int(g["n_tgt"]))
    pose, tup = headline["plain"]
    assert used == int(g["scales_used"])
    assert np.allclose(headline["runs"][0]["des_r"][:used], g["des_r"][:used], atol=1e-6)
    rs = int(g["row_stride"])
    report, flips = {}, 0
    acc_o, acc_g = [], []          # accumulated correspondences (scale, source keypoint, target keypoint) in accumulation order
    for ri in range(used):
        run = headline["runs"][ri]
        cap, scale = run["cap"], run["scale"]
        assert scale == ri
        m = int(_np(cap["counts"])[0])
        gs, gt = g[f"s{scale}_s_mids"], g[f"s{scale}_t_mids"]
        so, to = _np(cap["s_mids"])[:m], _np(cap["t_mids"])[:m]
        a = set(zip(so.tolist(), to.tolist()))
        b = set(zip(gs.tolist(), gt.tolist()))
        acc_o += [(scale, int(x), int(y)) for x, y in zip(so, to)]
        acc_g += [(scale, int(x), int(y)) for x, y in zip(gs, gt)]
        bad = 0
        for c, k in enumerate(("src", "tgt")):
            d = np.abs(_np(cap["desc"][c])[::rs].astype(np.float64) - g[f"s{scale}_{k}_desc"]).max(1)
            bad += int((d > 2e-5).sum())
        report[f"scale{scale}"] = dict(mutual_gpu=len(a), mutual_ref=len(b), mutual_differ=len(a ^ b), desc_rows_off=bad, desc_rows_checked=2 * len(d))
        flips += len(a ^ b)
        # a keypoint whose patch holds a point within an ulp of a radius / voxel bound has a descriptor that differs at the 1e-3 level
        # between the reference's torch / numpy arithmetic and the contract (the desc_rows_off rows): its match can differ.  Bounded at 3
        # per scale and counted (round 3's three pairs: 0); consensus set, RANSAC inliers and pose below are NOT relaxed
        assert len(a ^ b) <= 3, report
        # bounded, not just reported (advisor, round 3): at most 0.4 % of the sampled descriptor rows may sit beyond 2e-5
        assert bad <= 0.004 * 2 * len(d), report
        if a == b and np.array_equal(so, gs):
            report[f"scale{scale}"]["ind_max_diff"] = float(np.abs(_np(cap["ind"])[:m] - g[f"s{scale}_ind"]).max())
    k = 0
    while f"est{k}_T" in g:
        k += 1
    last = headline["runs"][used - 1]["cap"]
    C = int(_np(last["counts"])[2])
    gi = g[f"est{k - 1}_inlier_ind"]
    oi = _np(last["inlier_ind"])[:C]
    cons_o, cons_g = {acc_o[j] for j in oi.tolist()}, {acc_g[j] for j in gi.tolist()}
    report["consensus"] = dict(gpu=len(cons_o), ref=len(cons_g), common=len(cons_o & cons_g))
    report["matches_that_differ"] = flips
    rre, rte = bx.synth.pose_difference(pose, g["pose"])
    report["pose_diff_deg_m"] = (rre, rte)
    print("\nREALSIZE_REPORT", headline["name"], report)
    out = os.environ.get("BX_REALSIZE_REPORT")
    if out:
        import json
        with open(out, "a") as f:
            f.write(json.dumps({headline["name"]: report}) + "\n")
    assert flips <= PINNED_FLIPS[headline["name"]], report      # pinned per fixture (zero on five of the seven)
    assert (tup[0], tup[2]) == (int(g["num_inliers"]), int(g["num_inlier_ind"])) and abs(tup[1] - int(g["num_mutual"])) <= flips
    assert cons_o == cons_g, report                # the consensus set as correspondences
    if flips == 0:
        assert set(oi.tolist()) == set(gi.tolist()), report
    assert rre < 1e-4 and rte < 1e-4, report      # north_star tolerance


# ------------------------------------------------------------------ every conv layer beyond the persistent-walk threshold
DESC_SHAPES = [(3, 64), (4, 64), (4, 128), (8, 128), (8, 64), (4, 64), (4, 32), (2, 32)]


def test_conv_layers_group_walk(bx, packed, oracle, monkeypatch):
    """All 8 Cylindrical_Net layers at 5000 units and all 9 explicit CostNet layers at 1400 units, each run on the GPU output of
    the layer before it, with the persistent grid capped at 48 workgroups (BX_CONV_PERSIST_CAP) so that EVERY layer walks >= 2
    unit groups per workgroup (grp_next slab hand-off, k_conv.hip) -- plus the uncapped Desc layers, which walk at K = 5000 with
    the real occupancy.  NSAMP random units per layer vs the oracle fed the GPU's input of that layer."""
    import torch
    from bufferx_amd import lib
    W = bx.weights
    cfg = _headline_cfg(bx)
    rng = np.random.default_rng(3)
    for cap_env in ("48", None):
        if cap_env:
            monkeypatch.setenv("BX_CONV_PERSIST_CAP", cap_env)
        else:
            monkeypatch.delenv("BX_CONV_PERSIST_CAP", raising=False)
        ctx = lib.Context(cfg, max_points=4096, device=0, packed_weights=packed)
        try:
            # Desc: layer inputs are non-negative (post-ReLU) activations like the real ones
            units = K
            x = torch.as_tensor(np.abs(rng.standard_normal((units, 3, 140, 16))).astype(np.float32), device="cuda:0")
            tap = W.cyl_tap_table(cfg.patch.ele_n, cfg.patch.azi_n)
            for l, (L, (nch, cout)) in enumerate(zip(packed["desc"], DESC_SHAPES)):
                assert x.shape[1] == nch
                y = ctx.conv_layer(0, l, x, (units, (cout + 15) // 16, 140, 16))
                sel = np.sort(rng.choice(units, NSAMP, replace=False))
                sel[-1] = units - 1                                   # the ragged tail group
                ts = torch.as_tensor(sel, device=x.device)
                ref = oracle.desc_conv(lib.chunked_to_logical(_np(x[ts])), tap, L["W"], L["b"], L["relu"])
                assert np.array_equal(lib.chunked_to_logical(_np(y[ts])), ref), ("desc", l, cap_env)
                x = y
            if cap_env is None:
                continue
            # Pose: layer 0 consumes the implicit cost volume (bx_pose_net); layers 1..9 through bx_conv_layer
            units = 1400
            geo = W.pose_geometry(cfg.patch.ele_n, cfg.patch.azi_n)
            dims0 = geo[1][0]
            pin = int(np.prod(dims0))
            x = torch.as_tensor(np.abs(rng.standard_normal((units, 2, pin, 16))).astype(np.float32), device="cuda:0")
            for l in range(1, 10):
                L = packed["pose"][l]
                dims, k, out = geo[l]
                tap, _ = W.valid_tap_table(dims, k)
                cout = L["W"].shape[-1]
                y = ctx.conv_layer(1, l, x, (units, (cout + 15) // 16, int(np.prod(out)), 16))
                sel = np.sort(rng.choice(units, NSAMP, replace=False))
                sel[-1] = units - 1
                ts = torch.as_tensor(sel, device=x.device)
                ref = oracle.pose_conv(l, lib.chunked_to_logical(_np(x[ts])), tap, dims, L["W"], L["b"], L["relu"])
                assert np.array_equal(lib.chunked_to_logical(_np(y[ts])), ref), ("pose", l)
                x = y
        finally:
            ctx.close()


# ------------------------------------------------------------------ BASELINE configs[0]: one whole pair against the oracle
def test_config0_pair_bit_exact(bx, packed, oracle):
    """Single 3DMatch-like pair, 1 scale, 512 FPS keypoints, 512 points per patch, N ~ 30k, RANSAC + refinement: the whole
    pipeline bit-identical to the CPU oracle pipeline (BASELINE.json configs[0] is exactly this CPU run)."""
    from oracle import pipeline as PL
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 512, 512, 1
    cfg.patch.search_radius_thresholds = [5]
    pair = bx.synth.make_pair(0, "indoor", n_target=30000, shared=True)
    seed = 0
    ctx = lib.Context(cfg, max_points=max(len(pair["src"]), len(pair["tgt"])), device=0, packed_weights=packed)
    perm_s = np.stack([oracle.make_perm(len(pair["src"]), seed, 0)])
    perm_t = np.stack([oracle.make_perm(len(pair["tgt"]), seed, 1)])
    res = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed)
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed)
    ctx.close()
    assert (res.num_inliers, res.num_mutual, res.num_inlier_ind, res.scales_used) == tuple(ref[1:])
    assert np.array_equal(np.array(res.pose).reshape(4, 4), np.asarray(ref[0], np.float64))


# ------------------------------------------------------------------ > 200 000 points: the radius estimation's subsample
def test_cloud_above_200k_points(bx, packed, oracle):
    """models/BUFFERX.py:661-665: a cloud of more than 200 000 points is subsampled (with replacement) to 200 000 for the radius
    estimation while `num_pts` keeps the original size.  230k / 150k-point clouds, bit-exact against the oracle pipeline."""
    from oracle import pipeline as PL
    from bufferx_amd import lib
    cfg = bx.make_cfg("KITTI")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 96, 64, 2
    cfg.patch.search_radius_thresholds = [2, 0.5]
    cfg.patch.num_points_radius_estimate = 128
    cfg.match.iter_n = 800
    rng = np.random.default_rng(12)
    base = bx.synth.make_pair(9, "outdoor", voxel=0.05)
    # densify the sweep: jittered copies of the surface samples
    def dense(c, n):
        reps = int(np.ceil(n / len(c)))
        out = np.concatenate([c + rng.normal(0, 0.02, c.shape).astype(np.float32) for _ in range(reps)])[:n]
        return np.ascontiguousarray(out[rng.permutation(n)], np.float32)
    src, tgt = dense(base["src"], 230000), dense(base["tgt"], 150000)
    seed = 3
    ctx = lib.Context(cfg, max_points=len(src), device=0, packed_weights=packed)
    S2 = 2
    perm_s = np.stack([oracle.make_perm(len(src), seed, 2 * i) for i in range(S2)])
    perm_t = np.stack([oracle.make_perm(len(tgt), seed, 2 * i + 1) for i in range(S2)])
    res = ctx.register_pair(src, tgt, True, perm_s, perm_t, seed)
    ref = PL.register_pair(src, tgt, packed, cfg, True, seed)
    ctx.close()
    assert (res.num_inliers, res.num_mutual, res.num_inlier_ind, res.scales_used) == tuple(ref[1:])
    assert np.array_equal(np.array(res.pose).reshape(4, 4), np.asarray(ref[0], np.float64))


# ------------------------------------------------------------------ early exit armed but NOT taken, 3 scales
def test_early_exit_not_taken_three_scales(bx, packed, oracle):
    """enable_early_exit with a threshold the first scale cannot reach (models/BUFFERX.py:424-457): the RANSAC call of scale 0 runs,
    the exit is not taken, all three scales run and a second RANSAC call (next seed stream) produces the pose."""
    from oracle import pipeline as PL
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 200, 96, 3
    cfg.patch.search_radius_thresholds = [5, 2, 0.5]
    cfg.patch.num_points_radius_estimate = 200
    cfg.match.iter_n = 3000
    cfg.match.enable_early_exit = True
    cfg.match.early_exit_min_inliers = 100000
    pair = bx.synth.make_pair(14, "indoor", n_target=6000, identical=True)
    seed = 21
    ctx = lib.Context(cfg, max_points=len(pair["src"]), device=0, packed_weights=packed)
    perm_s = np.stack([oracle.make_perm(len(pair["src"]), seed, 2 * i) for i in range(3)])
    perm_t = np.stack([oracle.make_perm(len(pair["tgt"]), seed, 2 * i + 1) for i in range(3)])
    res = ctx.register_pair(pair["src"], pair["tgt"], pair["aligned_z"], perm_s, perm_t, seed)
    ref = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed)
    ctx.close()
    assert res.scales_used == 3 and ref[4] == 3
    assert (res.num_inliers, res.num_mutual, res.num_inlier_ind, res.scales_used) == tuple(ref[1:])
    assert np.array_equal(np.array(res.pose).reshape(4, 4), np.asarray(ref[0], np.float64))
