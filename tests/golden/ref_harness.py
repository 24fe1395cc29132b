"""tests/golden/ref_harness.py -- imports the REAL reference (/root/reference) on CPU.

Only usable in the build container (the reference tree is not on the GPU box); it is used by
tests/golden/make_golden.py to mint the committed fixtures and by tests that are skipped when
/root/reference is absent.

The reference's own Python (models/*.py, utils/common.py, utils/SE3.py) runs UNMODIFIED.  Its
un-vendored third-party CUDA/C++ dependencies are replaced by numpy/torch stubs that restate their
published semantics -- written independently of oracle/bx_oracle.c (vectorised
numpy, numpy SVD) so that oracle-vs-golden agreement is a real cross-check:
    pointnet2_ops.pointnet2_utils : furthest_point_sample, gather_operation, ball_query, grouping_operation
    knn_cuda.KNN, torch_batch_svd.svd, kornia.geometry.conversions.axis_angle_to_rotation_matrix,
    open3d (PointCloud / Vector*Vector / registration_ransac_based_on_correspondence), easydict.
Randomness the reference leaves unseeded is made explicit: np.random.choice inside select_patches is
served from a queue of permutations, the RANSAC stub uses the counter RNG of the C-ABI contract.
"""
import os
import sys
import types
import numpy as np
import torch

REF = "/root/reference"
MASK64 = (1 << 64) - 1


def mix64(seed, ctr):
    z = (seed + 0x9E3779B97F4A7C15 * (ctr + 1)) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return (z ^ (z >> 31)) & MASK64


def make_perm(n, seed, stream):
    keys = np.array([mix64(seed, (stream << 32) + i) for i in range(n)], dtype=np.uint64)
    return np.argsort(keys, kind="stable").astype(np.int64)


# ---------------------------------------------------------------- pointnet2_ops stubs
# BX_REF_DIST_FORM=nvcc_fma (round 5, a study switch): squared distances as nvcc's default -fmad=true contracts the upstream kernels'
# source expression, fmaf(dz, dz, fmaf(dy, dy, dx * dx)); default = the un-fused evaluation.  fmaf is emulated in binary64 (the product
# of two binary32 numbers is exact there; the one extra rounding of the sum can differ from a true fmaf only when the exact result lies
# within 2^-29 ulp of a binary32 tie -- the C oracle uses the hardware instruction, and the two are compared).
DIST_FORM = os.environ.get("BX_REF_DIST_FORM", "unfused")


def _fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _d2(dx, dy, dz):
    if DIST_FORM == "nvcc_fma":
        return _fma32(dz, dz, _fma32(dy, dy, dx * dx))
    return (dx * dx + dy * dy) + dz * dz


def _fps_np(xyz, m):
    xyz = np.asarray(xyz, np.float32)
    n = len(xyz)
    T = 1
    while T * 2 <= n and T * 2 <= 512:
        T *= 2
    temp = np.full(n, 1e10, np.float32)
    mag = _d2(xyz[:, 0], xyz[:, 1], xyz[:, 2])
    ok = ~(mag <= np.float32(1e-3))
    ar = np.arange(n)
    tid = ar % T
    idx = np.zeros(m, np.int32)
    old = 0
    for j in range(1, m):
        d = xyz - xyz[old]
        d = _d2(d[:, 0], d[:, 1], d[:, 2])
        temp = np.where(ok, np.minimum(d, temp), temp)
        if not ok.any():
            old = 0
        else:
            cand = np.where(ok, temp, np.float32(-1))
            mx = cand.max()
            c = ar[(cand == mx) & ok]
            old = int(c[np.lexsort((c, tid[c]))[0]])
        idx[j] = old
    return idx


def _ball_query_np(r, ns, xyz, new_xyz):
    xyz = np.asarray(xyz, np.float32)
    new_xyz = np.asarray(new_xyz, np.float32)
    r2 = np.float32(r) * np.float32(r)
    out = np.zeros((len(new_xyz), ns), np.int32)
    for j, q in enumerate(new_xyz):
        d = q - xyz
        d2 = _d2(d[:, 0], d[:, 1], d[:, 2])
        hit = np.flatnonzero(d2 < r2)[:ns]
        if len(hit):
            out[j, :] = hit[0]
            out[j, :len(hit)] = hit
    return out


class _Pnt2:
    calls = []

    @staticmethod
    def furthest_point_sample(xyz, m):
        return torch.from_numpy(np.stack([_fps_np(x.numpy(), m) for x in xyz]))

    @staticmethod
    def gather_operation(f, idx):
        return torch.stack([f[b][:, idx[b].long()] for b in range(f.shape[0])])

    @staticmethod
    def ball_query(r, ns, xyz, new_xyz):
        return torch.from_numpy(np.stack([_ball_query_np(r, ns, xyz[b].numpy(), new_xyz[b].numpy())
                                          for b in range(xyz.shape[0])]))

    @staticmethod
    def grouping_operation(f, idx):
        return torch.stack([f[b][:, idx[b].long()] for b in range(f.shape[0])])


class _KNN:
    def __init__(self, k=1, transpose_mode=True):
        assert k == 1 and transpose_mode

    def __call__(self, ref, query):
        r = ref[0].numpy().astype(np.float32)
        q = query[0].numpy().astype(np.float32)
        idx = np.zeros(len(q), np.int64)
        dist = np.zeros(len(q), np.float32)
        for i in range(0, len(q), 256):
            d = ((q[i:i + 256, None, :] - r[None, :, :]) ** 2).sum(-1)
            idx[i:i + 256] = d.argmin(1)
            dist[i:i + 256] = np.sqrt(d.min(1))
        return torch.from_numpy(dist)[None, :, None], torch.from_numpy(idx)[None, :, None]


def _svd(A):
    U, S, Vh = torch.linalg.svd(A)
    return U, S, Vh.transpose(-1, -2)


def _axis_angle_to_rotation_matrix(axis_angle):
    """kornia.geometry.conversions.axis_angle_to_rotation_matrix (kornia >= 0.7), restated."""
    def _normal(axis_angle, theta2, eps=1e-6):
        k_one = 1.0
        theta = torch.sqrt(theta2)
        wxyz = axis_angle / (theta + eps)
        wx, wy, wz = torch.chunk(wxyz, 3, dim=1)
        cos_theta = torch.cos(theta)
        sin_theta = torch.sin(theta)
        r00 = cos_theta + wx * wx * (k_one - cos_theta)
        r10 = wz * sin_theta + wx * wy * (k_one - cos_theta)
        r20 = -wy * sin_theta + wx * wz * (k_one - cos_theta)
        r01 = wx * wy * (k_one - cos_theta) - wz * sin_theta
        r11 = cos_theta + wy * wy * (k_one - cos_theta)
        r21 = wx * sin_theta + wy * wz * (k_one - cos_theta)
        r02 = wy * sin_theta + wx * wz * (k_one - cos_theta)
        r12 = -wx * sin_theta + wy * wz * (k_one - cos_theta)
        r22 = cos_theta + wz * wz * (k_one - cos_theta)
        return torch.cat([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)

    def _taylor(axis_angle):
        rx, ry, rz = torch.chunk(axis_angle, 3, dim=1)
        k_one = torch.ones_like(rx)
        return torch.cat([k_one, -rz, ry, rz, k_one, -rx, -ry, rx, k_one], dim=1).view(-1, 3, 3)

    _aa = torch.unsqueeze(axis_angle, dim=1)
    theta2 = torch.squeeze(torch.matmul(_aa, _aa.transpose(1, 2)), dim=1)
    rn = _normal(axis_angle, theta2)
    rt = _taylor(axis_angle)
    mask = (theta2 > 1e-6).view(-1, 1, 1)
    mp = mask.type_as(theta2)
    mn = (~mask).type_as(theta2)
    return mp * rn + mn * rt


# ---------------------------------------------------------------- open3d stub (RANSAC restated in numpy fp64)
class _Vec(list):
    def __init__(self, a):
        super().__init__()
        self.a = np.asarray(a)


class _PointCloud:
    def __init__(self):
        self.points = None
        self.colors = None


class _Result:
    def __init__(self):
        self.transformation = np.eye(4)
        self.correspondence_set = []
        self.fitness = 0.0
        self.inlier_rmse = 0.0


RANSAC_STATE = {"seed": 0, "calls": 0, "log": []}


def _umeyama(a, b):
    ma, mb = a.mean(0), b.mean(0)
    sigma = (b - mb).T @ (a - ma) / len(a)
    U, D, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[-1] = -1
    R = U @ np.diag(S) @ Vt
    return R, mb - R @ ma, D


def _ransac(pcd0, pcd1, corr, dist_th, estimation, ransac_n, checkers, criteria):
    src = np.asarray(pcd0.points.a, np.float64)
    tgt = np.asarray(pcd1.points.a, np.float64)
    corr = np.asarray(corr.a).reshape(-1, 2)
    C = len(corr)
    res = _Result()
    seed = mix64(RANSAC_STATE["seed"], 0x5AC0000 + RANSAC_STATE["calls"])
    RANSAC_STATE["calls"] += 1
    if C < 3:
        RANSAC_STATE["log"].append((C, 0, 0))
        return res
    sim = checkers[0].th
    best_inl, best_rmse, est_k = 0, 0.0, criteria.max_iteration
    s_all, t_all = src[corr[:, 0]], tgt[corr[:, 1]]
    itr = 0
    while itr < criteria.max_iteration and itr < est_k:
        sel = [mix64(seed, itr * 3 + j) % C for j in range(3)]
        itr += 1
        a, b = s_all[sel], t_all[sel]
        R, t, D = _umeyama(a, b)
        if not (D[1] > D[0] * 1e-12):   # rank-deficient sample: rejected (documented deviation)
            continue
        ok = True
        for i in range(3):
            for j in range(i + 1, 3):
                ds, dt = np.linalg.norm(a[i] - a[j]), np.linalg.norm(b[i] - b[j])
                if ds < dt * sim or dt < ds * sim:
                    ok = False
        if not ok:
            continue
        if (np.linalg.norm(b - (a @ R.T + t), axis=1) > dist_th).any():
            continue
        dis = np.linalg.norm(s_all @ R.T + t - t_all, axis=1)
        inl = dis < dist_th
        n = int(inl.sum())
        rmse = float(np.sqrt((dis[inl] ** 2).sum() / n)) if n else 0.0
        if n > best_inl or (n == best_inl and rmse < best_rmse):
            best_inl, best_rmse = n, rmse
            res.transformation = np.eye(4)
            res.transformation[:3, :3] = R
            res.transformation[:3, 3] = t
            res.correspondence_set = [tuple(c) for c in corr[inl]]
            if criteria.confidence < 1.0:
                ratio = n / C
                r3 = ratio ** 3
                est = 0.0 if r3 >= 1.0 else np.log(1.0 - criteria.confidence) / np.log(1.0 - r3)
                if est < est_k:
                    est_k = int(np.ceil(est))
    RANSAC_STATE["log"].append((C, itr, best_inl))
    return res


class _Checker:
    def __init__(self, th):
        self.th = th


class _Criteria:
    def __init__(self, max_iteration, confidence):
        self.max_iteration, self.confidence = max_iteration, confidence


def _install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    p = mod("pointnet2_ops")
    pu = mod("pointnet2_ops.pointnet2_utils")
    for k in ("furthest_point_sample", "gather_operation", "ball_query", "grouping_operation"):
        setattr(pu, k, getattr(_Pnt2, k))
    p.pointnet2_utils = pu
    mod("knn_cuda").KNN = _KNN
    mod("torch_batch_svd").svd = _svd
    k = mod("kornia")
    kg = mod("kornia.geometry")
    kc = mod("kornia.geometry.conversions")
    kc.axis_angle_to_rotation_matrix = _axis_angle_to_rotation_matrix
    k.geometry = kg
    kg.conversions = kc
    o = mod("open3d")
    o.geometry = types.SimpleNamespace(PointCloud=_PointCloud)
    o.utility = types.SimpleNamespace(Vector3dVector=_Vec, Vector2iVector=_Vec)
    reg = types.SimpleNamespace(
        registration_ransac_based_on_correspondence=_ransac,
        TransformationEstimationPointToPoint=lambda s=False: None,
        CorrespondenceCheckerBasedOnEdgeLength=_Checker,
        CorrespondenceCheckerBasedOnDistance=_Checker,
        RANSACConvergenceCriteria=_Criteria)
    o.pipelines = types.SimpleNamespace(registration=reg)

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for a, b in dict(d or {}, **kw).items():
                self[a] = b

        def __setitem__(self, a, b):
            if isinstance(b, dict) and not isinstance(b, EasyDict):
                b = EasyDict(b)
            super().__setitem__(a, b)

        __setattr__ = __setitem__

        def __getattr__(self, a):
            try:
                return self[a]
            except KeyError:
                raise AttributeError(a)

    mod("easydict").EasyDict = EasyDict
    for name in ("nibabel", "tensorboardX"):
        mod(name)


_LOADED = {}
PERM_QUEUE = []
PERM_LOG = []


def load_reference():
    """Returns a namespace with the reference modules (BUFFERX, patch_embedder, patchnet, common, config)."""
    if _LOADED:
        return _LOADED["ns"]
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self  # models/patch_embedder.py:158 hard-codes .cuda()
    import models.BUFFERX as BX
    import models.patch_embedder as PE
    import models.patchnet as PN
    import utils.common as CM
    import config as CFG

    _orig_choice = np.random.choice

    def _choice(n, size=None, replace=True, p=None):
        if PERM_QUEUE and size == n and replace is False:
            perm = PERM_QUEUE.pop(0)
            assert len(perm) == n
            PERM_LOG.append(perm)
            return perm
        return _orig_choice(n, size, replace, p)

    np.random.choice = _choice
    ns = types.SimpleNamespace(BX=BX, PE=PE, PN=PN, CM=CM, CFG=CFG)
    _LOADED["ns"] = ns
    return ns
