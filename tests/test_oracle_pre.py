"""CPU: the pre-processing oracle (oracle/pre_oracle.py) against the fixtures minted from the REAL
reference function utils/tools.py::sphericity_based_voxel_analysis + scikit-learn (tests/golden/make_golden_pre.py), and the
voxel down-sampling restatement against an independent dictionary-accumulation transcription of the published Open3D
algorithm."""
import os
import numpy as np
import pytest

from oracle import pre_oracle as PO

CASES = ["pre_indoor", "pre_outdoor", "pre_flat"]


@pytest.mark.parametrize("name", CASES)
def test_analysis_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    vs, sph, aligned = PO.sphericity_based_voxel_analysis(g["src"], g["tgt"], g["idx_src"], g["idx_tgt"])
    assert vs == float(g["voxel_size"])
    assert aligned == bool(g["aligned"])
    assert abs(sph - float(g["sphericity"])) <= 1e-9 * max(1.0, abs(float(g["sphericity"])))
    w, comp, mean = PO.pca_stats(g["src"], g["idx_src"])
    assert np.allclose(w, g["ev_src"], rtol=1e-10, atol=1e-13)
    assert np.allclose(comp, g["comp_src"], atol=1e-9)
    assert np.allclose(mean, g["mean_src"], rtol=1e-12, atol=1e-12)


def _voxel_down_sample_dict(pts, vs):
    """Open3D 0.18 PointCloud::VoxelDownSample transcribed literally (map of accumulators, sequential AddPoint)."""
    p = np.asarray(pts, np.float32).astype(np.float64)
    origin = p.min(0) - vs * 0.5
    acc = {}
    for i in range(len(p)):
        ref = (p[i] - origin) / vs
        key = (int(np.floor(ref[0])), int(np.floor(ref[1])), int(np.floor(ref[2])))
        if key not in acc:
            acc[key] = [np.zeros(3), 0]
        acc[key][0] = acc[key][0] + p[i]
        acc[key][1] += 1
    return np.array([s / float(c) for s, c in acc.values()]).astype(np.float32)   # dict = insertion (first appearance) order


@pytest.mark.parametrize("seed,n,vs", [(0, 3000, 0.05), (1, 5000, 0.2), (2, 800, 1.0), (3, 1, 0.1)])
def test_voxel_down_sample_restatement(seed, n, vs):
    rng = np.random.default_rng(seed)
    pts = (rng.random((n, 3), np.float32) * np.float32([3, 2, 1]) - 1).astype(np.float32)
    a = PO.voxel_down_sample(pts, vs)
    b = _voxel_down_sample_dict(pts, vs)
    assert a.shape == b.shape and np.array_equal(a, b)


def test_voxel_down_sample_properties():
    rng = np.random.default_rng(5)
    pts = rng.random((4000, 3), np.float32)
    out = PO.voxel_down_sample(pts, 0.1)
    assert 500 < len(out) <= 11 ** 3
    # idempotent up to the half-voxel origin shift: every centroid lies inside the cloud's bounding box
    assert (out >= pts.min(0) - 1e-6).all() and (out <= pts.max(0) + 1e-6).all()
    # a voxel size larger than the cloud collapses it to its mean
    one = PO.voxel_down_sample(pts, 10.0)
    assert one.shape == (1, 3) and np.allclose(one[0], pts.astype(np.float64).mean(0), atol=1e-6)
