"""oracle/pre_oracle.py -- TEST INFRASTRUCTURE (CPU restatement, NumPy): the step in front of the hot path
(SURVEY.md §8f rank 1).

  voxel_down_sample            open3d==0.18.0 PointCloud::VoxelDownSample as called at dataset/threedmatch.py:90-102,
                               dataset/kitti.py, dataset/tiers.py (un-vendored C++; published algorithm restated: voxel
                               origin = min_bound - voxel_size/2, index = floor((p - origin)/voxel_size) in binary64, a
                               voxel's point = sum of its members IN INPUT ORDER / count, binary64).  Open3D emits the
                               voxels in unordered_map order (unspecified; the loaders shuffle afterwards,
                               dataset/threedmatch.py:94,109); this restatement emits them in order of first
                               appearance.  PARITY UNPINNED: Open3D is not installable here and the reference has no
                               vectors for it.
  pca_stats / sphericity_based_voxel_analysis
                               utils/tools.py:132-198 (compute_pca_alignment, sphericity_based_voxel_analysis) with
                               sklearn.decomposition.PCA restated (covariance + eigh, v-based sign flip).  Pinned against
                               the REAL reference function + scikit-learn by tests/golden/pre_*.npz
                               (tests/golden/make_golden_pre.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import numpy as np


def voxel_down_sample(pts, voxel_size):
    """pts float32 [n,3] -> float32 [m,3] voxel centroids, first-appearance order."""
    p = np.asarray(pts, np.float32).astype(np.float64)
    vs = float(voxel_size)
    if len(p) == 0:
        return np.zeros((0, 3), np.float32)
    origin = p.min(0) - vs * 0.5
    vidx = np.floor((p - origin) / vs).astype(np.int64)
    _, first, inv = np.unique(vidx, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    m = len(first)
    sums = np.zeros((m, 3), np.float64)
    np.add.at(sums, inv, p)                       # sequential, i.e. input order inside every voxel
    cnt = np.bincount(inv, minlength=m).astype(np.float64)
    cen = sums / cnt[:, None]
    order = np.argsort(first, kind="stable")      # first appearance
    return cen[order].astype(np.float32)


def pca_stats(pts, sample_idx):
    """sklearn.decomposition.PCA(n_components=3).fit(pts[sample_idx]) restated.
    Returns (explained_variance[3] descending, components[3,3] rows, mean[3])."""
    x = np.asarray(pts, np.float32).astype(np.float64)[np.asarray(sample_idx)]
    n = len(x)
    mean = x.mean(0)
    c = x.T @ x
    c -= n * np.outer(mean, mean)
    c /= n - 1
    w, v = np.linalg.eigh(c)
    w = w[::-1].copy()
    comp = v[:, ::-1].T.copy()
    w[w < 0] = 0.0
    sgn = np.sign(comp[np.arange(3), np.argmax(np.abs(comp), axis=1)])
    sgn[sgn == 0] = 1.0
    return w, comp * sgn[:, None], mean


def compute_pca_alignment(pts, sample_idx):
    """utils/tools.py:132-149"""
    w, comp, mean = pca_stats(pts, sample_idx)
    l1, l2, l3 = sorted(w, reverse=True)
    sphericity = l3 / l1
    z = comp[-1] / np.linalg.norm(comp[-1])
    is_aligned = abs(float(np.dot(z, [0.0, 0.0, 1.0]))) > 0.98
    return sphericity, is_aligned, (w, comp, mean)


def sphericity_based_voxel_analysis(src, tgt, idx_src, idx_tgt):
    """utils/tools.py:152-198 with the two np.random.choice subsamples made explicit."""
    s_src, a_src, pca_src = compute_pca_alignment(src, idx_src)
    s_tgt, a_tgt, pca_tgt = compute_pca_alignment(tgt, idx_tgt)
    if len(src) > len(tgt):
        ref, sphericity, pca = src, s_src, pca_src
    else:
        ref, sphericity, pca = tgt, s_tgt, pca_tgt
    w, comp, mean = pca
    tz = (np.asarray(ref, np.float32).astype(np.float64) - mean) @ comp[2]
    z_range = tz.max() - tz.min()
    alpha = 1.0 if sphericity < 0.05 else 1.5
    voxel_size = max(np.sqrt(z_range) / 100 * alpha, 0.001)
    zs = pca_src[1][-1] / np.linalg.norm(pca_src[1][-1])
    zt = pca_tgt[1][-1] / np.linalg.norm(pca_tgt[1][-1])
    same = float(np.dot(zs, zt)) > 0.96
    return round(float(voxel_size), 4), float(sphericity), bool(a_src and a_tgt and same)


_M64 = (1 << 64) - 1


def _mix64(seed, ctr):
    """splitmix64 finaliser over seed + golden * (ctr + 1): buffer-x_amd/csrc/bx_common.h::bx_mix64 == oracle/bx_oracle.c::bxo_mix64"""
    z = (seed + 0x9E3779B97F4A7C15 * (ctr + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def random_perm(n, seed):
    """Restatement of bx_random_perm (k_pre.hip): 4-round Feistel over 2w bits + cycle walking.  Not a reference function: the
    reference draws its permutations from NumPy's global RNG; this is the device-side stand-in, specified here."""
    w = 1
    while (1 << (2 * w)) < n:
        w += 1
    mask = (1 << w) - 1

    def f(x):
        L, R = x >> w, x & mask
        for r in range(4):
            L, R = R, L ^ (_mix64(seed, (r << 32) | R) & mask)
        return (L << w) | R
    out = np.empty(n, np.int32)
    for i in range(n):
        x = f(i)
        while x >= n:
            x = f(x)
        out[i] = x
    return out
