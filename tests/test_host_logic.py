"""Host-side logic (CPU): config mirror, state-dict layout, BatchNorm folding + kernel packing + tap tables
checked against plain PyTorch fp32 convolutions, synthetic-data determinism, reference-compiled neighbour sets."""
import os
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F


def test_config_mirrors_reference_defaults(bx):
    c = bx.make_cfg("3DMatch")     # config/indoor_config.py:49-80 + threedmatch_config.py:12
    assert (c.patch.num_fps, c.patch.num_points_per_patch, c.patch.num_scales) == (1500, 512, 3)
    assert c.patch.search_radius_thresholds == [5, 2, 0.5] and c.patch.num_points_radius_estimate == 2000
    assert (c.match.dist_th, c.match.similar_th, c.match.confidence, c.match.iter_n) == (0.10, 0.8, 0.999, 50000)
    assert c.match.inlier_th == 1 / 3 and c.test.pose_refine is True and c.patch.is_aligned_to_global_z is False
    k = bx.make_cfg("KITTI")       # config/outdoor_config.py:49-82
    assert (k.match.dist_th, k.match.inlier_th, k.match.similar_th, k.match.confidence) == (0.30, 2.0, 0.9, 1.0)
    assert k.patch.is_aligned_to_global_z is True and k.test.pose_refine is False
    t = bx.make_cfg("TIERS_hetero")  # derives from the OUTDOOR base in the reference
    assert t.patch.is_aligned_to_global_z is True
    assert bx.make_cfg("ETH").match.inlier_th == 1.5
    with pytest.raises(ValueError):
        bx.make_cfg("nope")


def test_state_dict_layout_matches_reference_keys(bx):
    from bufferx_amd.model import BufferX
    cfg = bx.make_cfg("3DMatch")
    m = BufferX(cfg)
    spec = bx.weights.state_dict_spec()
    sd = m.state_dict()
    assert len(sd) == 105 == len(spec)
    for (k, v), (k2, shape) in zip(sd.items(), spec):
        assert k == k2 and tuple(v.shape) == tuple(shape), (k, k2)
    assert sum(v.numel() for v in sd.values()) == 909996
    # test.py:86-94 style loading: filter by stage substring, load non-strictly per stage
    syn = {k: torch.from_numpy(np.asarray(v)) for k, v in bx.weights.synthetic_state_dict(1).items()}
    for stage in ("Desc", "Pose"):
        part = {k: v for k, v in syn.items() if stage in k}
        new = m.state_dict()
        new.update(part)
        m.load_state_dict(new)
    for k, v in m.state_dict().items():
        assert torch.equal(v, syn[k]), k
    m.eval()
    nn.DataParallel(m)  # wrapping must work (test.py:105)
    with pytest.raises(RuntimeError):
        m({"src_fds_pcd": torch.zeros(10, 3), "tgt_fds_pcd": torch.zeros(10, 3), "is_aligned_to_global_z": False})


def _torch_modules(sd):
    from bufferx_amd.model import BufferX
    import bufferx_amd
    m = BufferX(bufferx_amd.make_cfg("3DMatch"))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.eval()


def _pad_cyl(x):
    """pad_image / pad_image_3d (reference utils/common.py:265-310): circular in W, zeros in H."""
    x = torch.cat([x[..., -1:], x, x[..., :1]], dim=-1)
    z = torch.zeros_like(x[..., :1, :])
    return torch.cat([z, x, z], dim=-2)


def test_fold_pack_and_taps_match_pytorch_conv(bx, oracle):
    """oracle.conv on folded+packed weights == plain PyTorch fp32 conv + eval BatchNorm (tolerance: fp32 reassociation)."""
    sd = bx.weights.synthetic_state_dict(2)
    pw = bx.weights.fold_and_pack(sd)
    m = _torch_modules(sd)
    rng = np.random.default_rng(0)
    K = 3
    x = np.abs(rng.standard_normal((K, 16, 3, 7, 20))).astype(np.float32)
    # torch reference: Cylindrical_Net forward (models/patchnet.py:49-67)
    with torch.no_grad():
        t = torch.from_numpy(x)
        ops = m.Desc.conv_net.ops
        t = F.relu(ops[1](ops[0](_pad_cyl(t)))).squeeze(2)
        for i in range(3, 21, 3):
            t = F.relu(ops[i + 1](ops[i](_pad_cyl(t))))
        t = ops[21](_pad_cyl(t))
    feat = np.ascontiguousarray(x.transpose(0, 2, 3, 4, 1).reshape(K, 3, 140, 16))   # [K][rad][pos][c]
    y = feat
    tap = bx.weights.cyl_tap_table()
    for L in pw["desc"]:
        y = oracle.conv(y, tap, L["W"], L["b"], L["relu"])
    got = y.reshape(K, 2, 7, 20, 16).transpose(0, 1, 4, 2, 3).reshape(K, 32, 7, 20)
    assert np.abs(got - t.numpy()).max() < 2e-4 * max(1.0, np.abs(t.numpy()).max())
    # CostNet on a random cost volume
    c = rng.standard_normal((2, 32, 20, 5, 20)).astype(np.float32)
    with torch.no_grad():
        t = torch.from_numpy(c)
        for op in m.Pose.conv.ops:
            t = op(t)
    y = np.ascontiguousarray(c.reshape(2, 2, 16, 2000).transpose(0, 1, 3, 2))
    for L, (dims, k, _) in zip(pw["pose"], bx.weights.pose_geometry()):
        tp, _ = bx.weights.valid_tap_table(dims, k)
        y = oracle.conv(y, tp, L["W"], L["b"], L["relu"])
    got = y.reshape(2, 32)[:, :20]
    assert np.abs(got - t.numpy().reshape(2, 20)).max() < 5e-4 * max(1.0, np.abs(t.numpy()).max())


def test_pnt_and_pool_fold_match_pytorch(bx, oracle):
    sd = bx.weights.synthetic_state_dict(3)
    pw = bx.weights.fold_and_pack(sd)
    m = _torch_modules(sd)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 32, 7, 20)).astype(np.float32)
    with torch.no_grad():
        t = torch.from_numpy(x)
        w = m.Desc.pool_layer(t)
        f = F.normalize(F.avg_pool2d(t * w, kernel_size=(7, 20)).view(4, -1), p=2, dim=1)
        e = F.normalize(t, p=2, dim=1)
    xc = np.ascontiguousarray(x.reshape(4, 2, 16, 140).transpose(0, 1, 3, 2))
    desc, equi = oracle.desc_head(xc, pw["pool_w1"], pw["pool_b1"], pw["pool_w2"], pw["pool_b2"])
    assert np.abs(desc - f.numpy()).max() < 1e-5
    assert np.abs(equi.reshape(4, 7, 20, 32).transpose(0, 3, 1, 2) - e.numpy()).max() < 1e-5


def test_synth_is_deterministic(bx):
    a = bx.synth.make_pair(5, "indoor", n_target=2000)
    b = bx.synth.make_pair(5, "indoor", n_target=2000)
    assert np.array_equal(a["src"], b["src"]) and np.array_equal(a["tgt"], b["tgt"]) and np.array_equal(a["T_gt"], b["T_gt"])
    c = bx.synth.make_pair(6, "indoor", n_target=2000, identical=True)
    assert np.abs(c["src"].astype(np.float64) @ c["T_gt"][:3, :3].T + c["T_gt"][:3, 3] - c["tgt"]).max() < 1e-6
    o = bx.synth.make_pair(1, "outdoor", voxel=0.6)
    assert o["aligned_z"] is True and len(o["src"]) > 1000


def test_neighbour_sets_match_reference_nanoflann(oracle, bx):
    """cpp_wrappers radius search compiled from the reference tree (oracle/_ref): same neighbour SETS as the oracle's
    brute force, and ball_query's first-P-in-index-order list == the P smallest indices of that set."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    pair = bx.synth.make_pair(9, "indoor", n_target=4000)
    pts = pair["src"]
    q = pts[oracle.fps(pts, 64)]
    r = 0.22
    cnt = oracle.radius_counts(q, pts, np.float32(r))
    out, rcnt = oracle.ref_radius_neighbors(q, pts, np.float32(r), int(cnt.max()))
    assert np.array_equal(cnt, rcnt)
    P = 48
    idx, _ = oracle.ball_group(pts, q, np.float32(r), P)
    for i in range(len(q)):
        s = np.sort(out[i, :cnt[i]])
        assert len(np.unique(s)) == cnt[i]
        k = min(P, cnt[i])
        assert np.array_equal(idx[i, :k], s[:k])
        assert (idx[i, k:] == idx[i, 0]).all()


def test_bench_workloads_generate_valid_pairs(bx):
    """bench.py's synthetic workloads (BASELINE configs[1..4]) on CPU: every generator returns finite float32 clouds, a rigid ground
    truth, per-scale permutations that are permutations, and is deterministic in its seed (every rank builds the same pair list)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert set(bench.WORKLOADS) >= {"3dmatch", "3dlomatch", "3dmatch-noisy", "kitti", "tiers"}
    for wl in ("3dmatch", "3dlomatch", "kitti", "tiers"):
        a = bench.make_inputs(bx, 1, 100, 3, wl)[0]
        b = bench.make_inputs(bx, 1, 100, 3, wl)[0]
        for k in ("src", "tgt"):
            assert a[k].dtype == np.float32 and a[k].ndim == 2 and a[k].shape[1] == 3 and np.isfinite(a[k]).all()
            assert np.array_equal(a[k], b[k])
        T = np.asarray(a["T_gt"], np.float64)
        R = T[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1.0) < 1e-6
        for k, n in (("perm_src", len(a["src"])), ("perm_tgt", len(a["tgt"]))):
            assert a[k].shape == (3, n) and a[k].dtype == np.int32
            assert all(np.array_equal(np.sort(row), np.arange(n)) for row in a[k])
            assert np.array_equal(a[k], b[k])
    # the registering workload: the two fragments share surface samples inside the overlap (what lets random weights register it)
    p = bench.make_inputs(bx, 1, 100, 3, "3dmatch")[0]
    moved = (p["src"].astype(np.float64) @ np.asarray(p["T_gt"])[:3, :3].T + np.asarray(p["T_gt"])[:3, 3]).astype(np.float32)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(p["tgt"]).query(moved)
    assert (d < 1e-4).mean() > 0.3
