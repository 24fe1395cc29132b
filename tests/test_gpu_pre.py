"""GPU parity of the pre-processing entry points (bx_pre_*) through the C-ABI: voxel down-sampling is
bit-exact against the oracle (binary64 accumulation in input order), the PCA statistics agree to 1e-9 (binary64 reductions in a
different order) and the reference function's outputs (voxel size rounded to 4 decimals, z-alignment flag) are reproduced
exactly for the fixtures minted from the real reference code."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pctx(bx, packed):
    from bufferx_amd import lib
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 64, 32, 1
    cfg.patch.search_radius_thresholds = [2]
    c = lib.Context(cfg, max_points=1024, device=0, packed_weights=packed)
    c.pre_reserve(400000)
    yield c
    c.close()


@pytest.mark.parametrize("seed,n,vs", [(0, 3000, 0.05), (1, 50000, 0.03), (2, 800, 1.0), (3, 1, 0.1), (4, 120000, 0.011),
                                       (5, 5000, 7.5)])
def test_voxel_downsample_exact(pctx, seed, n, vs):
    from oracle import pre_oracle as PO
    rng = np.random.default_rng(seed)
    pts = (rng.random((n, 3), np.float32) * np.float32([3, 2, 1]) - 1).astype(np.float32)
    if seed == 1:   # duplicates and points on voxel faces
        pts[1000:2000] = pts[:1000]
        pts[3000:3500] = np.round(pts[3000:3500] / np.float32(vs)) * np.float32(vs)
    out, cnt = pctx.pre_voxel_downsample(pts, vs)
    m, status = (int(v) for v in cnt.cpu().numpy())
    ref = PO.voxel_down_sample(pts, vs)
    assert status == 0 and m == len(ref)
    assert np.array_equal(out[:m].cpu().numpy(), ref)


def test_voxel_downsample_crowded_voxels(pctx):
    """Voxels with tens of thousands of members (voxel size large against the density): the per-voxel reduction switches from
    sorting its members to an index-order walk; same binary64 summation order, bounded work."""
    from oracle import pre_oracle as PO
    rng = np.random.default_rng(8)
    pts = np.concatenate([rng.random((60000, 3), np.float32) * np.float32(0.9),                 # one voxel with 60 k points
                          rng.random((3000, 3), np.float32) * 4 + np.float32([2, 0, 0]),         # sparse part
                          np.tile(np.float32([[5.5, 5.5, 5.5]]), (700, 1))]).astype(np.float32)  # 700 identical points
    pts = pts[rng.permutation(len(pts))]
    for vs in (1.0, 0.25):
        out, cnt = pctx.pre_voxel_downsample(pts, vs)
        m, status = (int(v) for v in cnt.cpu().numpy())
        ref = PO.voxel_down_sample(pts, vs)
        assert status == 0 and m == len(ref) and np.array_equal(out[:m].cpu().numpy(), ref)


def test_voxel_downsample_real_like_cloud(pctx, bx):
    from oracle import pre_oracle as PO
    pair = bx.synth.make_pair(11, "indoor", n_target=30000, voxel=0.008)   # dense "raw" fragment
    pts = pair["src"]
    out, cnt = pctx.pre_voxel_downsample(pts, 0.025)
    m = int(cnt.cpu().numpy()[0])
    ref = PO.voxel_down_sample(pts, 0.025)
    assert m == len(ref) and np.array_equal(out[:m].cpu().numpy(), ref)


def test_voxel_size_too_small_is_reported(pctx):
    rng = np.random.default_rng(0)
    pts = (rng.random((4000, 3)) * 100).astype(np.float32)
    out, cnt = pctx.pre_voxel_downsample(pts, 1e-5)      # 10^7 voxels along an axis > 2^21
    assert tuple(int(v) for v in cnt.cpu().numpy()) == (0, 1)


def test_voxel_downsample_large_extent(pctx, bx):
    """outdoor sweep at centimetre voxels: ~10^10 grid cells, the hash table keeps memory O(n)"""
    from oracle import pre_oracle as PO
    pts = np.ascontiguousarray(bx.synth.make_pair(3, "outdoor", voxel=0.1)["src"], np.float32)
    out, cnt = pctx.pre_voxel_downsample(pts, 0.02)
    m, status = (int(v) for v in cnt.cpu().numpy())
    ref = PO.voxel_down_sample(pts, 0.02)
    assert status == 0 and m == len(ref) and np.array_equal(out[:m].cpu().numpy(), ref)


@pytest.mark.parametrize("name", ["pre_indoor", "pre_outdoor", "pre_flat"])
def test_pca_and_analysis_match_reference(pctx, golden_dir, name):
    from oracle import pre_oracle as PO
    from bufferx_amd.preprocess import Preprocessor
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    st = pctx.pre_pca(g["src"], g["idx_src"]).cpu().numpy()
    w, comp, mean = PO.pca_stats(g["src"], g["idx_src"])
    assert np.allclose(st[0:3], w, rtol=1e-9, atol=1e-12)
    assert np.allclose(st[3:12].reshape(3, 3), comp, atol=1e-8)
    assert np.allclose(st[12:15], mean, rtol=1e-12, atol=1e-12)
    assert np.allclose(st[0:3], g["ev_src"], rtol=1e-9, atol=1e-12)          # scikit-learn itself
    tz = (g["src"].astype(np.float64) - mean) @ comp[2]
    assert np.isclose(st[15], tz.min(), rtol=1e-9, atol=1e-9) and np.isclose(st[16], tz.max(), rtol=1e-9, atol=1e-9)
    # the mirror of the reference function, with the reference's own RNG calls replayed
    seeds = {"pre_indoor": 7, "pre_outdoor": 3, "pre_flat": 11}
    np.random.seed(seeds[name])
    pre = Preprocessor(pctx, 400000)
    vs, sph, aligned = pre.sphericity_based_voxel_analysis(g["src"], g["tgt"])
    assert vs == float(g["voxel_size"]) and aligned == bool(g["aligned"])
    assert abs(sph - float(g["sphericity"])) <= 1e-8 * max(1.0, abs(float(g["sphericity"])))
    # and the down-sampling at that voxel size
    ds = pre.voxel_down_sample(g["src"], vs)
    assert np.array_equal(ds.cpu().numpy(), PO.voxel_down_sample(g["src"], vs))


@pytest.mark.parametrize("n", [1, 2, 5, 64, 1000, 4097, 65536])
def test_random_perm_matches_oracle(pctx, n):
    from oracle import pre_oracle as PO
    for seed in (0, 12345678901234567):
        got = pctx.random_perm(n, seed).cpu().numpy()
        if n <= 4097:
            assert np.array_equal(got, PO.random_perm(n, seed))
        assert np.array_equal(np.sort(got), np.arange(n))
    a, b = pctx.random_perm(max(n, 2), 1).cpu().numpy(), pctx.random_perm(max(n, 2), 2).cpu().numpy()
    assert n < 64 or not np.array_equal(a, b)


def test_voxel_downsample_hand_computed(pctx):
    """the hand-computed Open3D VoxelDownSample vector of tests/test_oracle_pre.py (points on voxel faces, negative coordinates,
    min-bound offset) through bx_pre_voxel_downsample"""
    from test_oracle_pre import VOXEL_KAT_PTS, VOXEL_KAT_OUT
    out, cnt = pctx.pre_voxel_downsample(VOXEL_KAT_PTS, 0.5)
    m, status = (int(v) for v in cnt.cpu().numpy())
    assert status == 0 and m == len(VOXEL_KAT_OUT)
    assert np.array_equal(out[:m].cpu().numpy(), VOXEL_KAT_OUT)
