"""Seeded synthetic registration pairs (no datasets exist offline).

Clouds are float32 [N,3], voxel-deduplicated (barycentre per voxel, like Open3D voxel_down_sample
used by the reference loaders, dataset/threedmatch.py:90-109) and shuffled.  tgt = T_gt * crop(scene)
(+ jitter), so T_gt maps src -> tgt like the reference's `relt_pose`.
"""
import numpy as np


def _voxel_down(pts, voxel, rng):
    key = np.floor(pts / voxel).astype(np.int64)
    key -= key.min(0)
    dims = key.max(0) + 1
    lin = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    order = np.argsort(lin, kind="stable")
    lin_s = lin[order]
    start = np.flatnonzero(np.r_[True, lin_s[1:] != lin_s[:-1]])
    cnt = np.diff(np.r_[start, len(lin_s)])
    out = np.add.reduceat(pts[order].astype(np.float64), start, axis=0) / cnt[:, None]
    out = out.astype(np.float32)
    rng.shuffle(out)
    return out


def _rot(axis, ang):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _sample_box_surface(rng, lo, hi, n):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    ext = hi - lo
    areas = np.array([ext[1] * ext[2], ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[2], ext[0] * ext[1], ext[0] * ext[1]])
    face = rng.choice(6, n, p=areas / areas.sum())
    p = lo + rng.random((n, 3)) * ext
    ax = face // 2
    side = face % 2
    p[np.arange(n), ax] = np.where(side == 0, lo[ax], hi[ax])
    return p


def indoor_scene(rng, n_raw, extent=(3.0, 3.0, 2.5), n_boxes=8):
    """Room-like surface sample: 6 bounding planes + boxes (furniture)."""
    ext = np.asarray(extent)
    parts = [_sample_box_surface(rng, -ext / 2, ext / 2, n_raw // 2)]
    per = (n_raw - n_raw // 2) // n_boxes
    for _ in range(n_boxes):
        size = rng.uniform(0.25, 0.9, 3) * np.array([1, 1, 0.8])
        c = rng.uniform(-ext / 2 + size / 2, ext / 2 - size / 2)
        c[2] = -ext[2] / 2 + size[2] / 2
        parts.append(_sample_box_surface(rng, c - size / 2, c + size / 2, per))
    return np.concatenate(parts)


def outdoor_scene(rng, rings=64, az_steps=2048, max_range=80.0, n_boxes=40, sensor_h=1.73, box_range=(6.0, 60.0), box_size=(1.5, 12.0, 9.0),
                  elev_deg=(-24.8, 2.0), return_rings=False):
    """LiDAR-like sweep: ring pattern over a ground plane and vertical boxes (buildings, cars)."""
    elev = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], rings))
    az = np.linspace(0, 2 * np.pi, az_steps, endpoint=False)
    E, A = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    t = np.full(len(d), np.inf)
    down = d[:, 2] < -1e-6
    t[down] = -sensor_h / -d[down, 2] * -1.0
    t[down] = sensor_h / (-d[down, 2])
    for _ in range(n_boxes):
        size = np.array([rng.uniform(box_size[0], box_size[1]), rng.uniform(box_size[0], box_size[1]), rng.uniform(1.4, box_size[2])])
        r = rng.uniform(box_range[0], box_range[1])
        th = rng.uniform(0, 2 * np.pi)
        c = np.array([r * np.cos(th), r * np.sin(th), -sensor_h + size[2] / 2])
        lo, hi = c - size / 2, c + size / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = lo / d
            t2 = hi / d
        tn = np.minimum(t1, t2).max(1)
        tf = np.maximum(t1, t2).min(1)
        hit = (tn <= tf) & (tn > 0.5)
        t = np.where(hit & (tn < t), tn, t)
    ok = np.isfinite(t) & (t < max_range)
    pts = d[ok] * t[ok, None]
    pts += rng.normal(0, 0.01, pts.shape)
    if return_rings:
        return pts, (np.arange(len(d)) // az_steps)[ok]
    return pts


def make_pair(seed, kind="indoor", n_target=30000, overlap=0.6, voxel=None, max_rot_deg=None, max_trans=None,
              jitter=None, identical=False, shared=False):
    """Returns dict(src, tgt, T_gt (4x4 float64, src->tgt), aligned_z).

    shared=True (indoor): noise-free PARTIAL-overlap pair whose two fragments are crops of ONE voxelised sample of the scene,
    shuffled independently: inside the overlap the surface samples coincide (tgt = fp32(R p + t)), outside they do not exist in
    the other fragment.  Keypoints are chosen by each fragment's own FPS; the ones that coincide (about K^2/N of them) carry
    identical neighbourhoods, which is what lets a randomly initialised network register the pair."""
    rng = np.random.default_rng(seed)
    if shared:
        assert kind == "indoor"
        voxel = 0.025 if voxel is None else voxel
        scene = _voxel_down(indoor_scene(rng, int(n_target * 4.5)), voxel, rng).astype(np.float64)
        half = 1.5
        a = half * overlap / (2.0 - overlap)
        R = _rot(rng.normal(size=3), np.deg2rad(rng.uniform(10, 45 if max_rot_deg is None else max_rot_deg)))
        t = rng.uniform(-1, 1, 3) * (0.8 if max_trans is None else max_trans)
        off = np.array([0.3, -0.2, 1.6])
        src = scene[scene[:, 0] < a] + off
        tgt = (scene[scene[:, 0] > -a] + off) @ R.T + t
        # thin both fragments to n_target points with ONE keep-mask over the scene, so that the overlap keeps coinciding samples
        keep = rng.permutation(len(scene))
        rank = np.empty(len(scene), np.int64)
        rank[keep] = np.arange(len(scene))
        rs, rt = rank[scene[:, 0] < a], rank[scene[:, 0] > -a]
        m = max(len(src), len(tgt))
        if n_target and m > n_target:
            cut = int(len(scene) * n_target / m)
            src, tgt = src[rs < cut], tgt[rt < cut]
        src = src[rng.permutation(len(src))]
        tgt = tgt[rng.permutation(len(tgt))]
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = t
        return dict(src=np.ascontiguousarray(src, np.float32), tgt=np.ascontiguousarray(tgt, np.float32), T_gt=T, aligned_z=False)
    if kind == "indoor":
        voxel = 0.025 if voxel is None else voxel
        jitter = 0.002 if jitter is None else jitter
        scene = indoor_scene(rng, int(n_target * 4.5))
        half = 1.5
        # src sees x < a, tgt sees x > -a  -> overlap fraction a/ (a + half) roughly
        a = half * overlap / (2.0 - overlap)
        src = scene[scene[:, 0] < a]
        tgt = scene[scene[:, 0] > -a]
        src = src[rng.random(len(src)) < 0.7]
        tgt = tgt[rng.random(len(tgt)) < 0.7]
        R = _rot(rng.normal(size=3), np.deg2rad(rng.uniform(10, 45 if max_rot_deg is None else max_rot_deg)))
        t = rng.uniform(-1, 1, 3) * (0.8 if max_trans is None else max_trans)
        aligned = False
        # move both clouds away from the origin like real scans (camera frame)
        off = np.array([0.3, -0.2, 1.6])
        src = src + off
        tgt = tgt + off
    elif kind == "outdoor":
        voxel = 0.30 if voxel is None else voxel
        jitter = 0.01 if jitter is None else jitter
        scene = outdoor_scene(rng)
        # second sweep from a displaced, yaw-rotated pose (planar motion -> z axis shared)
        yaw = np.deg2rad(rng.uniform(-10, 10) if max_rot_deg is None else rng.uniform(-max_rot_deg, max_rot_deg))
        R = _rot([0, 0, 1], yaw)
        t = np.array([rng.uniform(3, 8) if max_trans is None else max_trans, rng.uniform(-0.5, 0.5), 0.0])
        src = scene[rng.random(len(scene)) < 0.85]
        tgt = scene[rng.random(len(scene)) < 0.85]
        aligned = True
    else:
        raise ValueError(kind)
    if identical:
        # noise-free, full-overlap pair with identical point order: tgt_i = R src_i + t
        src = _voxel_down(src, voxel, rng)
        if n_target and kind == "indoor":
            src = src[:n_target]
        src = src.astype(np.float32)
        tgt = (src.astype(np.float64) @ R.T + t).astype(np.float32)
    else:
        tgt = tgt + rng.uniform(-jitter, jitter, tgt.shape)
        tgt = tgt @ R.T + t
        src = _voxel_down(src, voxel, rng)
        tgt = _voxel_down(tgt, voxel, rng)
        if n_target and kind == "indoor":
            src = src[:n_target]
            tgt = tgt[:n_target]
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return dict(src=np.ascontiguousarray(src, np.float32), tgt=np.ascontiguousarray(tgt, np.float32), T_gt=T,
                aligned_z=aligned)


def make_tiers_pair(seed):
    """TIERS_hetero-like pair (dense -> sparse LiDAR at indoor-scale range, z axes aligned): src = one 128-ring
    sweep (~100k points, range <= 20 m), tgt = the 64 even rings of the SAME sweep seen from a displaced, yaw-rotated pose
    (tgt = fp32(R p + t), noise-free), shuffled independently.  The samples of the sparse sensor coincide with samples of the dense
    one, which is what lets randomly initialised weights find enough consistent matches for the early exit to be exercised."""
    rng = np.random.default_rng(seed)
    pts, ring = outdoor_scene(rng, rings=128, az_steps=1024, max_range=20.0, n_boxes=30, box_range=(2.5, 16.0), box_size=(0.6, 4.0, 3.0),
                              elev_deg=(-45.0, 20.0), return_rings=True)
    yaw = np.deg2rad(rng.uniform(-25, 25))
    R = _rot([0, 0, 1], yaw)
    t = np.array([rng.uniform(0.5, 2.0), rng.uniform(-0.5, 0.5), 0.0])
    src = pts[rng.permutation(len(pts))]
    sparse = pts[ring % 2 == 0]
    tgt = (sparse @ R.T + t)[rng.permutation(len(sparse))]
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return dict(src=np.ascontiguousarray(src, np.float32), tgt=np.ascontiguousarray(tgt, np.float32), T_gt=T, aligned_z=True)


def pose_difference(T_a, T_b):
    """(rotation difference in degrees, translation difference in metres) of two NEARLY EQUAL poses, well-conditioned at zero:
    angle = 2 asin(|R_a - R_b|_F / (2 sqrt 2)).  The reference's RRE (arccos of the trace, `pose_error` below) cannot resolve
    differences below ~0.05 deg between float32 poses -- the trace of a float32 rotation carries ~1e-6 of rounding and
    arccos(1 - 1e-6 / 2) = 0.04 deg -- so parity of two implementations' poses is measured with this instead."""
    Ra, Rb = np.asarray(T_a, np.float64)[:3, :3], np.asarray(T_b, np.float64)[:3, :3]
    fro = float(np.linalg.norm(Ra - Rb))
    ang = 2.0 * np.degrees(np.arcsin(min(1.0, fro / (2.0 * np.sqrt(2.0)))))
    return float(ang), float(np.linalg.norm(np.asarray(T_a, np.float64)[:3, 3] - np.asarray(T_b, np.float64)[:3, 3]))


def pose_error(T_est, T_gt):
    """(RRE degrees, RTE metres), reference utils/SE3.py:134-165."""
    R = T_est[:3, :3] @ T_gt[:3, :3].T
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    return float(np.degrees(np.arccos(c))), float(np.linalg.norm(T_est[:3, 3] - T_gt[:3, 3]))
