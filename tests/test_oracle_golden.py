"""Pins the CPU oracle against fixtures minted from the REAL reference code (tests/golden/make_golden.py runs
BufferX.forward of /root/reference on CPU with numpy stubs for the un-vendored CUDA ops)."""
import os
import numpy as np
import pytest

from test_gpu_pipeline import CASES, MID, make_case


@pytest.fixture(scope="module")
def helpers(golden_dir):
    return np.load(os.path.join(golden_dir, "helpers.npz"))


def test_voxel_tables_bit_exact(oracle, helpers):
    cen, rot = oracle.voxel_table()
    assert np.array_equal(cen, helpers["voxel_centres"])          # get_voxel_coordinate (utils/common.py:422-428)
    R = helpers["inv_rot"]                                         # var_to_invar table (utils/common.py:483-493)
    assert np.array_equal(rot[:, 0], R[:, 0, 0]) and np.array_equal(rot[:, 1], R[:, 0, 1])
    assert np.array_equal(rot[:, 2], R[:, 1, 0]) and np.array_equal(rot[:, 3], R[:, 1, 1])
    # SURVEY.md Appendix C known answers
    assert np.allclose(cen[0], [0.03663022, 0.00580166, 0.16248799], atol=1e-8)
    assert np.allclose(cen[419], [0.18315111, -0.02900829, -0.81243993], atol=1e-8)


def test_radius_known_answers(oracle, helpers):
    """density_aware_radius_estimation on torch.rand(20000,3)*4 -> 1.01 / 0.72 / 0.44."""
    import torch
    torch.manual_seed(0)
    P = (torch.rand(20000, 3) * 4).numpy()
    assert np.array_equal(P[::50], helpers["radius_kat_pts"])
    got = [oracle.radius(P, len(P), P[:2000], t) for t in (5, 2, 0.5)]
    assert got == helpers["radius_kat"].tolist() == [1.01, 0.72, 0.44]


def test_axis_helpers_close_to_reference(oracle, helpers, packed):
    # RodsRotatFormula + cal_Z_axis are exercised through bxo_patch_features: build patches whose last point
    # is the centre and compare the returned R with the reference's for the same z axis.
    d, ref = helpers["calz_in"], helpers["calz_ref"]
    patches = np.concatenate([d + ref[:, None, :], ref[:, None, :]], 1).astype(np.float32)  # last slot = centre
    R, _ = oracle.patch_features(patches, 1.0, False, packed["pnt_w"], packed["pnt_b"])
    import torch
    z = torch.from_numpy(helpers["calz_out"])
    z = z / z.norm(dim=1, keepdim=True)
    # reference R for that z (restated RodsRotatFormula output is in the fixture for random z)
    zin, Rref = helpers["rods_in"], helpers["rods_out"]
    assert zin.shape == (64, 3) and Rref.shape == (64, 3, 3)
    # the oracle's R maps z -> +z axis:  z @ R == (0,0,1)
    zz = np.einsum("ki,kij->kj", z.numpy(), R.reshape(-1, 3, 3))
    assert np.allclose(zz, np.tile([0, 0, 1], (len(zz), 1)), atol=2e-6)
    zr = np.einsum("ki,kij->kj", zin, Rref)
    assert np.allclose(zr, np.tile([0, 0, 1], (64, 1)), atol=2e-6)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_pipeline_matches_reference(bx, packed, golden_dir, name):
    from oracle import pipeline as PL
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    rs = int(g["row_stride"]) if "row_stride" in g else 1          # the first fixtures (round 1) keep every row and every 16th map
    st = int(g["sub_stride"]) if "sub_stride" in g else 16
    cfg, pair, seed = make_case(bx, name)
    assert np.array_equal(pair["src"][:8], g["src_head"]) and len(pair["src"]) == int(g["n_src"])
    cap = {}
    pose, n_inl, n_mut, n_ind, scales = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed, cap)
    assert scales == int(g["scales_used"])
    # BASELINE configs[0] size (512 keypoints x 512 points on 18k-point clouds): a handful of patches hold a point whose distance sits
    # within an ulp of the radius / voxel bound, where the numpy stand-ins of the un-vendored CUDA ops (ref_harness.py) and the
    # oracle's arithmetic contract may decide differently (4 of 1024 descriptor rows differ at the 1e-3 level; LABBOOK.md section 4).
    # There: >= 99 % of the rows within the strict bound and every row within 1e-2; the small cases stay strict for every row.
    big = name == "baseline_cfg0" or name in MID

    def close(a, b, tol, axis_rows=True):
        d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
        if not big:
            return d.max() < tol
        rows = d.reshape(d.shape[0], -1).max(1)
        return (rows < tol).mean() >= 0.99 and rows.max() < 1e-2

    flips = 0
    for i in range(scales):
        assert cap[f"s{i}_des_r"] == pytest.approx(float(g["des_r"][i]), abs=1e-12)
        for c in ("src", "tgt"):
            tag = f"s{i}_{c}_"
            assert close(cap[tag + "desc"][::rs], g[tag + "desc"], 2e-5)
            assert close(cap[tag + "R"].reshape(-1, 3, 3)[::rs], g[tag + "R"], 1e-5)
            equi = cap[tag + "equi"].reshape(-1, 7, 20, 32).transpose(0, 3, 1, 2)[::st]
            assert close(equi, g[tag + "equi_sub"], 1e-5)
        a = set(zip(cap[f"s{i}_s_mids"].tolist(), cap[f"s{i}_t_mids"].tolist()))
        b = set(zip(g[f"s{i}_s_mids"].tolist(), g[f"s{i}_t_mids"].tolist()))
        flips += len(a ^ b)
        if name in MID:
            # a keypoint whose patch holds a point within an ulp of the radius / a voxel bound has a descriptor that differs at the 1e-3
            # level between the reference's torch / numpy arithmetic and the contract (the rows the `close` budget above allows): its
            # match can differ.  Bounded and counted; consensus set, RANSAC inliers and pose below are NOT relaxed.
            assert len(a ^ b) <= 3, (name, i, sorted(a ^ b))
        else:
            assert np.array_equal(cap[f"s{i}_s_mids"], g[f"s{i}_s_mids"]) and np.array_equal(cap[f"s{i}_t_mids"], g[f"s{i}_t_mids"])
        if a == b:
            assert close(cap[f"s{i}_ind"], g[f"s{i}_ind"], 5e-5)
    k = 0
    while f"est{k}_T" in g:
        k += 1
    # the consensus set as CORRESPONDENCES (scale, source keypoint, target keypoint): an index into the accumulated arrays shifts when a
    # match in front of it differs
    acc_o = [(i, int(x), int(y)) for i in range(scales) for x, y in zip(cap[f"s{i}_s_mids"], cap[f"s{i}_t_mids"])]
    acc_g = [(i, int(x), int(y)) for i in range(scales) for x, y in zip(g[f"s{i}_s_mids"], g[f"s{i}_t_mids"])]
    assert {acc_o[j] for j in cap[f"s{scales - 1}_inlier_ind"]} == {acc_g[j] for j in g[f"est{k - 1}_inlier_ind"]}
    if flips == 0:
        assert np.array_equal(cap[f"s{scales - 1}_inlier_ind"], g[f"est{k - 1}_inlier_ind"])
        assert np.abs(cap["init_pose"] - g[f"est{k - 1}_T"]).max() < 1e-9
    assert (n_inl, n_ind) == (int(g["num_inliers"]), int(g["num_inlier_ind"])) and abs(n_mut - int(g["num_mutual"])) <= flips
    rre, rte = bx.synth.pose_difference(np.asarray(pose, np.float64), g["pose"])   # well-conditioned at zero (synth.py)
    if flips:
        print("\nMID_FLIPS", name, flips, "of", len(acc_g), "pose diff", rre, rte)
    assert rre < 1e-4 and rte < 1e-4      # north_star tolerance: 1e-4 deg / 1e-4 m


# ------------------------------------------------------------------ the real-size fixtures (K = 5000 / P = 1024 / S = 3)
# smallest fraction of the sampled descriptor rows of one (scale, cloud) within 2e-5 of the reference's, per fixture: the z-aligned
# configurations agree in every row; the un-aligned (indoor) ones have a few rows per thousand with a point within an ulp of a radius /
# voxel bound (LABBOOK.md section 4)
BIG_ROWS_MIN_FRAC = {"headline_cfg1": 0.997, "kitti_cfg2": 1.0, "tiers_early": 1.0, "headline_cfg1_b": 0.995, "headline_cfg1_c": 0.997,
                     "kitti_cfg2_b": 1.0, "headline_lo": 0.996}     # observed 0.9976 / 1 / 1 / 0.9952 / 0.9976 / 1 / 0.9968 (of 1 250 rows)
BIG_NAMES = ["headline_cfg1", "kitti_cfg2", "tiers_early", "headline_cfg1_b", "headline_cfg1_c", "kitti_cfg2_b", "headline_lo"]


@pytest.mark.skipif(not os.environ.get("BX_RUN_BIG_ORACLE"), reason="3-4 CPU-minutes per case: set BX_RUN_BIG_ORACLE=1 (results quoted in LABBOOK.md section 4)")
@pytest.mark.parametrize("name", BIG_NAMES)
def test_oracle_matches_reference_at_real_size(bx, packed, golden_dir, name):
    """The CPU oracle pipeline against the fixture minted by the reference's own forward at BASELINE configs[1] / [2] / [4] size: radii,
    RANSAC inliers and consensus count identical; per-scale mutual sets as (source keypoint, target keypoint) correspondences with at most
    3 differing per scale (a keypoint whose patch holds a point within an ulp of a radius / voxel bound: the reference's torch / numpy
    arithmetic and the contract may decide differently -- zero on five of the seven fixtures); consensus set identical as correspondences;
    pose within 1e-4 deg / 1e-4 m; >= 99 % of the sampled descriptor rows of every (scale, cloud) within 2e-5.  (The GPU twin of this test runs in every
    `pytest -m gpu`: tests/test_gpu_headline.py::test_headline_vs_reference.)"""
    from oracle import pipeline as PL
    from test_gpu_headline import big_case, golden_path, PINNED_FLIPS
    g = np.load(golden_path(name))
    cfg, pair, seed = big_case(bx, name)
    cap = {}
    pose, n_inl, n_mut, n_ind, scales = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed, cap)
    assert scales == int(g["scales_used"])
    rs = int(g["row_stride"])
    flips, worst_rows = 0, 1.0
    for i in range(scales):
        assert cap[f"s{i}_des_r"] == pytest.approx(float(g["des_r"][i]), abs=1e-12)
        for c in ("src", "tgt"):
            d = np.abs(cap[f"s{i}_{c}_desc"][::rs].astype(np.float64) - g[f"s{i}_{c}_desc"]).max(1)
            worst_rows = min(worst_rows, float((d < 2e-5).mean()))
            # pinned per fixture just below what was observed (profiles/r05_big_oracle.jsonl: desc_rows_within_2e5_min_frac)
            assert (d < 2e-5).mean() >= BIG_ROWS_MIN_FRAC[name]
        a_ = set(zip(cap[f"s{i}_s_mids"].tolist(), cap[f"s{i}_t_mids"].tolist()))
        b_ = set(zip(g[f"s{i}_s_mids"].tolist(), g[f"s{i}_t_mids"].tolist()))
        assert len(a_ ^ b_) <= 3, (name, i, sorted(a_ ^ b_))
        flips += len(a_ ^ b_)
    k = 0
    while f"est{k}_T" in g:
        k += 1
    acc_o = [(i, int(x), int(y)) for i in range(scales) for x, y in zip(cap[f"s{i}_s_mids"], cap[f"s{i}_t_mids"])]
    acc_g = [(i, int(x), int(y)) for i in range(scales) for x, y in zip(g[f"s{i}_s_mids"], g[f"s{i}_t_mids"])]
    assert {acc_o[j] for j in cap[f"s{scales - 1}_inlier_ind"]} == {acc_g[j] for j in g[f"est{k - 1}_inlier_ind"]}
    if flips == 0:
        assert np.array_equal(cap[f"s{scales - 1}_inlier_ind"], g[f"est{k - 1}_inlier_ind"])
    assert (n_inl, n_ind) == (int(g["num_inliers"]), int(g["num_inlier_ind"])) and abs(n_mut - int(g["num_mutual"])) <= flips
    rre, rte = bx.synth.pose_difference(np.asarray(pose, np.float64), g["pose"])
    print("\nBIG_ORACLE", name, "flips", flips, "pose diff", rre, rte)
    out = os.environ.get("BX_BIG_ORACLE_REPORT")
    if out:
        import json
        with open(out, "a") as f:
            f.write(json.dumps(dict(case=name, matches_that_differ=flips, pinned=PINNED_FLIPS[name], desc_rows_within_2e5_min_frac=worst_rows,
                                    counts=[n_inl, n_mut, n_ind, scales], pose_diff_deg_m=[rre, rte])) + "\n")
    assert flips <= PINNED_FLIPS[name]
    assert rre < 1e-4 and rte < 1e-4
