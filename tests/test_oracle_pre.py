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


# ---------------------------------------------------------------------------------------------------- hand-computed known answer
# Open3D 0.18 PointCloud::VoxelDownSample: origin = min_bound - voxel/2, index = floor((p - origin) / voxel), output = mean of the
# members.  Every number below is a multiple of 1/16 (exact in binary32 / binary64), so the expected centroids were worked out by
# hand: voxel = 0.5, min_bound = (-1.25, -0.5, -0.25), origin = (-1.5, -0.75, -0.5);
#   p0 (-1.25, 0, 0)        -> (0.5, 1.5, 1.0)   -> voxel (0,1,1)
#   p1 (-1.0, 0.25, 0)      -> (1.0, 2.0, 1.0)   -> voxel (1,2,1)      x and y exactly ON a voxel face: the upper voxel
#   p2 (-0.75, 0, 0.25)     -> (1.5, 1.5, 1.5)   -> voxel (1,1,1)
#   p3 (-0.5, 0, 0)         -> (2.0, 1.5, 1.0)   -> voxel (2,1,1)      on a face again
#   p4 (1, 1, 1)            -> (5.0, 3.5, 3.0)   -> voxel (5,3,3)
#   p5 (1.125, 1.125, 1)    -> (5.25, 3.75, 3.0) -> voxel (5,3,3)      joins p4
#   p6 (-1.25, -0.5, -0.25) -> (0.5, 0.5, 0.5)   -> voxel (0,0,0)      the min-bound corner: half a voxel inside the grid
#   p7 (-0.875, .125, .125) -> (1.25, 1.75, 1.25)-> voxel (1,1,1)      joins p2
VOXEL_KAT_PTS = np.array([(-1.25, 0, 0), (-1.0, 0.25, 0), (-0.75, 0, 0.25), (-0.5, 0, 0), (1, 1, 1), (1.125, 1.125, 1),
                          (-1.25, -0.5, -0.25), (-0.875, 0.125, 0.125)], np.float32)
VOXEL_KAT_OUT = np.array([(-1.25, 0, 0), (-1.0, 0.25, 0), (-0.8125, 0.0625, 0.1875), (-0.5, 0, 0), (1.0625, 1.0625, 1.0),
                          (-1.25, -0.5, -0.25)], np.float32)          # voxels in order of first appearance


def test_voxel_down_sample_hand_computed():
    from oracle import pre_oracle as PO
    out = PO.voxel_down_sample(VOXEL_KAT_PTS, 0.5)
    assert np.array_equal(out, VOXEL_KAT_OUT)
    # as a set it does not depend on the emission order (Open3D's is that of an unordered_map)
    shuffled = VOXEL_KAT_PTS[[6, 3, 5, 0, 7, 1, 4, 2]]
    got = PO.voxel_down_sample(shuffled, 0.5)
    assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, VOXEL_KAT_OUT.tolist()))
