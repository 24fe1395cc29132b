"""ctypes binding of libbufferx_hip.so (C-ABI: include/bufferx.h).

PyTorch is used only as plumbing: device allocations (`tensor.data_ptr()`), the current HIP stream and
weight loading.  There is NO fallback: if the HIP library cannot be loaded every entry point raises.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("BX_HIP_SO") or os.path.join(_CSRC, "libbufferx_hip.so")   # BX_HIP_SO: kernel-experiment builds (tools/)
_LIB = None

BX_MAX_SCALES = 8


class BxParams(C.Structure):
    _fields_ = [("num_fps", C.c_int32), ("num_points_per_patch", C.c_int32), ("num_scales", C.c_int32),
                ("rad_n", C.c_int32), ("azi_n", C.c_int32), ("ele_n", C.c_int32), ("voxel_sample", C.c_int32),
                ("num_points_radius_estimate", C.c_int32), ("delta", C.c_double),
                ("search_radius_thresholds", C.c_double * BX_MAX_SCALES),
                ("dist_th", C.c_double), ("inlier_th", C.c_double), ("similar_th", C.c_double),
                ("confidence", C.c_double), ("iter_n", C.c_int32), ("enable_early_exit", C.c_int32),
                ("early_exit_min_inliers", C.c_int32), ("pose_refine", C.c_int32), ("max_points", C.c_int32),
                ("pose_estimator", C.c_int32), ("kiss_resolution", C.c_double),
                ("keypoint_tiles", C.c_int32), ("desc_conv_form", C.c_int32), ("pose_conv_form", C.c_int32),
                ("cost_l0_form", C.c_int32)]


class BxWeights(C.Structure):
    _fields_ = [("pnt_w", C.c_void_p), ("pnt_b", C.c_void_p), ("pool_w1", C.c_void_p), ("pool_b1", C.c_void_p),
                ("pool_w2", C.c_void_p), ("pool_b2", C.c_void_p), ("desc_w", C.c_void_p * 8), ("desc_b", C.c_void_p * 8),
                ("pose_w", C.c_void_p * 10), ("pose_b", C.c_void_p * 10)]


class BxResult(C.Structure):
    _fields_ = [("pose", C.c_double * 16), ("num_inliers", C.c_int32), ("num_mutual", C.c_int32),
                ("num_inlier_ind", C.c_int32), ("scales_used", C.c_int32), ("ransac_iters", C.c_int32),
                ("refine_iters", C.c_int32), ("status", C.c_int32), ("arith_forms", C.c_int32),
                ("des_r", C.c_float * BX_MAX_SCALES)]


class BxCapture(C.Structure):
    """include/bufferx.h bx_capture: caller-owned device buffers that receive the intermediates of one scale."""
    _fields_ = [("scale", C.c_int32), ("cloud", C.c_int32), ("pts_perm", C.c_void_p), ("patches", C.c_void_p),
                ("feat", C.c_void_p), ("x", C.c_void_p), ("kpts", C.c_void_p * 2), ("desc", C.c_void_p * 2),
                ("equi", C.c_void_p * 2), ("R", C.c_void_p * 2), ("s_mids", C.c_void_p), ("t_mids", C.c_void_p),
                ("ind", C.c_void_p), ("R_cat", C.c_void_p), ("t_cat", C.c_void_p), ("ss_cat", C.c_void_p),
                ("tt_cat", C.c_void_p), ("cons_cnt", C.c_void_p), ("inlier_ind", C.c_void_p), ("counts", C.c_void_p),
                ("T_ransac", C.c_void_p)]


EXPORTS = ["bx_create", "bx_destroy", "bx_last_error", "bx_load_weights", "bx_workspace_bytes", "bx_register_pair",
           "bx_register_pair_begin", "bx_register_pair_finish",
           "bx_set_capture", "bx_keypoint_tile_bounds",
           "bx_profile_enable", "bx_profile_read", "bx_debug_read",
           "bx_fps", "bx_radius", "bx_permute", "bx_ball_group", "bx_patch_features", "bx_ball_group_counted", "bx_patch_features_counted",
           "bx_desc_net", "bx_conv_layer",
           "bx_mutual", "bx_pose_net", "bx_hypotheses", "bx_consensus", "bx_ransac", "bx_kiss_solve", "bx_refine",
           "bx_pre_reserve", "bx_pre_voxel_downsample", "bx_pre_pca", "bx_random_perm",
           "bx_lane_create", "bx_lane_destroy", "bx_attach_lane",
           "bx_io_probe", "bx_io_read_xyz", "bx_prefetch_create", "bx_prefetch_submit", "bx_prefetch_wait", "bx_prefetch_release",
           "bx_prefetch_destroy"]


def build(force=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "bufferx.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale or force:
        subprocess.check_call(["make", "-C", _CSRC, "-j8"], stdout=subprocess.DEVNULL)
    return _SO


def load():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is None:
        # torch first: it ships its own HIP runtime; if /opt/rocm's copy were loaded before it (this library links
        # against libamdhip64), the two runtimes would disagree about the devices ("no ROCm-capable device")
        import torch  # noqa: F401
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP path has no CPU fallback)")
        lib = C.CDLL(_SO)
        lib.bx_last_error.restype = C.c_char_p
        lib.bx_workspace_bytes.restype = C.c_int64
        _LIB = lib
    return _LIB


class BxError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        msg = load().bx_last_error().decode(errors="replace")       # a message may quote bytes of a corrupt input file
        raise BxError(f"{what} failed (status {rc}): {msg}")


def params_from_cfg(cfg, max_points):
    p = BxParams()
    p.num_fps = int(cfg.patch.num_fps)
    p.num_points_per_patch = int(cfg.patch.num_points_per_patch)
    p.num_scales = int(cfg.patch.num_scales)
    p.rad_n, p.azi_n, p.ele_n = int(cfg.patch.rad_n), int(cfg.patch.azi_n), int(cfg.patch.ele_n)
    p.voxel_sample = int(cfg.patch.voxel_sample)
    p.num_points_radius_estimate = int(cfg.patch.num_points_radius_estimate)
    p.delta = float(cfg.patch.delta)
    thr = list(cfg.patch.search_radius_thresholds)
    assert len(thr) == p.num_scales, f"num_scales {p.num_scales} != num_thresholds {len(thr)}"  # models/BUFFERX.py:276
    for i, t in enumerate(thr):
        p.search_radius_thresholds[i] = float(t)
    p.dist_th, p.inlier_th = float(cfg.match.dist_th), float(cfg.match.inlier_th)
    p.similar_th, p.confidence = float(cfg.match.similar_th), float(cfg.match.confidence)
    p.iter_n = int(cfg.match.iter_n)
    p.enable_early_exit = int(bool(cfg.match.get("enable_early_exit", True)))
    p.early_exit_min_inliers = int(cfg.match.get("early_exit_min_inliers", 15))
    p.pose_refine = int(cfg.test.pose_refine is True)
    p.max_points = int(max_points)
    est = cfg.match.get("pose_estimator", "ransac")
    if est not in ("ransac", "kiss_matcher"):
        raise ValueError(f"Unknown pose estimator: {est}")           # models/pose_estimator.py:48
    p.pose_estimator = 1 if est == "kiss_matcher" else 0
    p.kiss_resolution = float(cfg.match.get("kiss_resolution", 0.3))
    # not a reference option: 2..8 = latency form of the whole-pair call (FPS beside the descriptor work), see include/bufferx.h
    p.keypoint_tiles = int(cfg.test.get("keypoint_tiles", 0))
    # arithmetic forms of the stages that have more than one (include/bufferx.h): cfg.arith, names as in config.ARITH_FORMS
    from . import config as _config
    ar = _config.arith_of(cfg)
    p.desc_conv_form = _config.ARITH_FORMS["desc_conv"].index(ar["desc_conv"])
    p.pose_conv_form = _config.ARITH_FORMS["pose_conv"].index(ar["pose_conv"])
    p.cost_l0_form = _config.ARITH_FORMS["cost_l0"].index(ar["cost_l0"])
    return p


def forms_of_result(res):
    """bx_result.arith_forms -> {"desc_conv": name, "pose_conv": name, "cost_l0": name}"""
    from . import config as _config
    v = int(res.arith_forms)
    return {"desc_conv": _config.ARITH_FORMS["desc_conv"][v & 255], "pose_conv": _config.ARITH_FORMS["pose_conv"][(v >> 8) & 255],
            "cost_l0": _config.ARITH_FORMS["cost_l0"][(v >> 16) & 255]}


def slot_perm():
    """slot -> logical channel map of one 16-chunk (include/bufferx.h bx_chunk_slot)."""
    inv = np.zeros(16, np.int64)
    for c in range(16):
        inv[4 * (c % 4) + c // 4] = c
    return inv


def chunked_to_logical(x):
    """[..., 16 slots] -> [..., 16 logical channels]"""
    inv = slot_perm()
    out = np.empty_like(x)
    out[..., inv] = x
    return out


def logical_to_chunked(x):
    inv = slot_perm()
    return np.ascontiguousarray(x[..., inv])


class Lane:
    """bx_lane: the contexts attached to one Lane run their MFMA-bound sections one after the other (mode 1: everything after
    the FPS; mode 2: the convolution stacks)."""

    def __init__(self, mode=2):
        self.lib = load()
        self.handle = C.c_void_p()
        _chk(self.lib.bx_lane_create(C.c_int32(int(mode)), C.byref(self.handle)), "bx_lane_create")

    def close(self):
        if self.handle:
            self.lib.bx_lane_destroy(self.handle)
            self.handle = C.c_void_p()


class Context:
    """One per in-flight pair: owns the device workspace arena (include/bufferx.h bx_ctx)."""

    def __init__(self, cfg, max_points, device=0, packed_weights=None):
        import torch
        self.torch = torch
        self.lib = load()
        self.cfg = cfg
        self.device = device
        self.params = params_from_cfg(cfg, max_points)
        self.handle = C.c_void_p()
        _chk(self.lib.bx_create(C.c_int(device), C.byref(self.params), C.byref(self.handle)), "bx_create")
        self._keep = []
        if packed_weights is not None:
            self.load_weights(packed_weights)

    def attach_lane(self, lane):
        """lane: Lane or None (see include/bufferx.h: ordering of the pairs in flight on one GPU)"""
        _chk(self.lib.bx_attach_lane(self.handle, lane.handle if lane is not None else C.c_void_p()), "bx_attach_lane")
        self._lane = lane

    def close(self):
        if self.handle:
            self.lib.bx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self.lib.bx_workspace_bytes(self.handle))

    def load_weights(self, pw):
        w = BxWeights()
        keep = []

        def ptr(a):
            a = np.ascontiguousarray(a, np.float32)
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)

        w.pnt_w, w.pnt_b = ptr(pw["pnt_w"]), ptr(pw["pnt_b"])
        w.pool_w1, w.pool_b1 = ptr(pw["pool_w1"]), ptr(pw["pool_b1"])
        w.pool_w2, w.pool_b2 = ptr(pw["pool_w2"]), ptr(pw["pool_b2"])
        for i, L in enumerate(pw["desc"]):
            w.desc_w[i], w.desc_b[i] = ptr(L["W"]), ptr(L["b"])
        for i, L in enumerate(pw["pose"]):
            w.pose_w[i], w.pose_b[i] = ptr(L["W"]), ptr(L["b"])
        _chk(self.lib.bx_load_weights(self.handle, C.byref(w)), "bx_load_weights")

    PROF_TAGS = ["fps", "radius", "neighbour_gather", "patch_features", "desc_conv", "desc_head", "mutual", "pose_net",
                 "consensus", "ransac", "refine", "permute", "neighbour_gather_query_kernel", "neighbour_grid_build"]

    def profile_enable(self, on=True):
        _chk(self.lib.bx_profile_enable(self.handle, C.c_int32(int(on))), "bx_profile_enable")

    def profile_read(self):
        """{stage: (total_ms, launches)} since the last read; call after synchronising the stream."""
        ms = (C.c_double * 16)()
        cnt = (C.c_int32 * 16)()
        _chk(self.lib.bx_profile_read(self.handle, ms, cnt), "bx_profile_read")
        return {t: (ms[i], cnt[i]) for i, t in enumerate(self.PROF_TAGS)}

    # ---------------------------------------------------------------- helpers
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype):
        t = self.torch
        if isinstance(a, t.Tensor):
            return a.to(device=f"cuda:{self.device}", dtype=dtype).contiguous()
        return t.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=f"cuda:{self.device}")

    def _empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    # ---------------------------------------------------------------- stages (torch tensors in / out)
    def fps(self, xyz, m):
        t = self.torch
        xyz = self._dev(xyz, t.float32)
        idx = self._empty((m,), t.int32)
        kp = self._empty((m, 3), t.float32)
        _chk(self.lib.bx_fps(self.handle, self._stream(), self._p(xyz), C.c_int32(xyz.shape[0]), C.c_int32(m), self._p(idx),
                             self._p(kp)), "bx_fps")
        return idx, kp

    def radius(self, pts, n_orig, kpts, thresholds):
        t = self.torch
        pts, kpts = self._dev(pts, t.float32), self._dev(kpts, t.float32)
        thr = (C.c_double * len(thresholds))(*[float(v) for v in thresholds])
        out = self._empty((len(thresholds),), t.float64)
        _chk(self.lib.bx_radius(self.handle, self._stream(), self._p(pts), C.c_int32(pts.shape[0]), C.c_int64(n_orig),
                                self._p(kpts), C.c_int32(kpts.shape[0]), thr, C.c_int32(len(thresholds)), self._p(out)),
             "bx_radius")
        return out

    def permute(self, pts, perm, out=None):
        t = self.torch
        pts, perm = self._dev(pts, t.float32), self._dev(perm, t.int32)
        out = self._empty(tuple(pts.shape), t.float32) if out is None else out[:pts.shape[0]]
        _chk(self.lib.bx_permute(self.handle, self._stream(), self._p(pts), self._p(perm), C.c_int32(pts.shape[0]),
                                 self._p(out)), "bx_permute")
        return out

    def ball_group(self, pts_perm, kpts, radius_dev, P, want_idx=True):
        t = self.torch
        pts_perm, kpts = self._dev(pts_perm, t.float32), self._dev(kpts, t.float32)
        radius_dev = self._dev(radius_dev, t.float64).reshape(-1)
        K = kpts.shape[0]
        idx = self._empty((K, P), t.int32) if want_idx else None
        patches = self._empty((K, P, 3), t.float32)
        _chk(self.lib.bx_ball_group(self.handle, self._stream(), self._p(pts_perm), C.c_int32(pts_perm.shape[0]), self._p(kpts),
                                    C.c_int32(K), self._p(radius_dev), C.c_int32(P), self._p(idx), self._p(patches)),
             "bx_ball_group")
        return idx, patches

    def ball_group_counted(self, pts_perm, kpts, radius_dev, P, fill=None):
        """the counted form (bx_ball_group_counted): -> (patches [K, P, 3] of which only the first counts[k] slots of row k are
        written -- the rest keeps `fill` (NaN by default, so that a consumer that reads beyond the count is caught) --, counts int32 [K])"""
        t = self.torch
        pts_perm, kpts = self._dev(pts_perm, t.float32), self._dev(kpts, t.float32)
        radius_dev = self._dev(radius_dev, t.float64).reshape(-1)
        K = kpts.shape[0]
        patches = t.full((K, P, 3), float("nan") if fill is None else fill, dtype=t.float32, device=f"cuda:{self.device}")
        counts = self._empty((K,), t.int32)
        _chk(self.lib.bx_ball_group_counted(self.handle, self._stream(), self._p(pts_perm), C.c_int32(pts_perm.shape[0]), self._p(kpts),
                                            C.c_int32(K), self._p(radius_dev), C.c_int32(P), self._p(patches), self._p(counts)),
             "bx_ball_group_counted")
        return patches, counts

    def patch_features_counted(self, patches, counts, kpts, radius_dev, aligned):
        t = self.torch
        patches, kpts = self._dev(patches, t.float32), self._dev(kpts, t.float32)
        counts = self._dev(counts, t.int32)
        radius_dev = self._dev(radius_dev, t.float64).reshape(-1)
        K, P, _ = patches.shape
        R = self._empty((K, 9), t.float32)
        feat = self._empty((K, 3, 140, 16), t.float32)
        _chk(self.lib.bx_patch_features_counted(self.handle, self._stream(), self._p(patches), self._p(counts), self._p(kpts), C.c_int32(K),
                                                C.c_int32(P), self._p(radius_dev), C.c_int32(int(aligned)), self._p(R), self._p(feat)),
             "bx_patch_features_counted")
        return R, feat

    # ---------------------------------------------------------------- pre-processing
    def pre_reserve(self, max_points):
        _chk(self.lib.bx_pre_reserve(self.handle, C.c_int64(int(max_points))), "bx_pre_reserve")

    def pre_voxel_downsample(self, pts, voxel_size, out=None, cnt=None):
        """-> (out float32 [n,3] device tensor (first m rows valid), count device int32[2] = {m, status}); `out` (>= n rows) and
        `cnt` may be caller-owned buffers (no allocation on the hot loop)"""
        t = self.torch
        pts = self._dev(pts, t.float32)
        n = pts.shape[0]
        out = self._empty((n, 3), t.float32) if out is None else out
        cnt = self._empty((2,), t.int32) if cnt is None else cnt
        _chk(self.lib.bx_pre_voxel_downsample(self.handle, self._stream(), self._p(pts), C.c_int32(n), C.c_double(float(voxel_size)),
                                              self._p(out), self._p(cnt)), "bx_pre_voxel_downsample")
        return out, cnt

    def random_perm(self, n, seed, out=None):
        """device int32 [n]: permutation of range(n) determined by seed (no host RNG, no H2D copy)"""
        out = self._empty((int(n),), self.torch.int32) if out is None else out[:int(n)]
        _chk(self.lib.bx_random_perm(self.handle, self._stream(), C.c_int32(int(n)), C.c_uint64(int(seed) & (2**64 - 1)), self._p(out)),
             "bx_random_perm")
        return out

    def pre_pca(self, pts, sample_idx, out=None):
        t = self.torch
        pts = self._dev(pts, t.float32)
        idx = self._dev(sample_idx, t.int32)
        out = self._empty((17,), t.float64) if out is None else out
        _chk(self.lib.bx_pre_pca(self.handle, self._stream(), self._p(pts), C.c_int32(pts.shape[0]), self._p(idx),
                                 C.c_int32(idx.shape[0]), self._p(out)), "bx_pre_pca")
        return out

    def patch_features(self, patches, radius_dev, aligned):
        t = self.torch
        patches = self._dev(patches, t.float32)
        radius_dev = self._dev(radius_dev, t.float64).reshape(-1)
        K, P, _ = patches.shape
        R = self._empty((K, 9), t.float32)
        feat = self._empty((K, 3, 140, 16), t.float32)
        _chk(self.lib.bx_patch_features(self.handle, self._stream(), self._p(patches), C.c_int32(K), C.c_int32(P),
                                        self._p(radius_dev), C.c_int32(int(aligned)), self._p(R), self._p(feat)),
             "bx_patch_features")
        return R, feat

    def desc_net(self, feat, want_x=False):
        t = self.torch
        feat = self._dev(feat, t.float32)
        K = feat.shape[0]
        desc = self._empty((K, 32), t.float32)
        equi = self._empty((K, 140, 32), t.float32)
        x = self._empty((K, 2, 140, 16), t.float32) if want_x else None
        _chk(self.lib.bx_desc_net(self.handle, self._stream(), self._p(feat), C.c_int32(K), self._p(desc), self._p(equi),
                                  self._p(x)), "bx_desc_net")
        return desc, equi, x

    def conv_layer(self, net, layer, x, out_shape):
        t = self.torch
        x = self._dev(x, t.float32)
        out = self._empty(tuple(out_shape), t.float32)
        _chk(self.lib.bx_conv_layer(self.handle, self._stream(), C.c_int32(net), C.c_int32(layer), self._p(x),
                                    C.c_int32(x.shape[0]), self._p(out)), "bx_conv_layer")
        return out

    def mutual(self, src_des, tgt_des):
        t = self.torch
        s, g = self._dev(src_des, t.float32), self._dev(tgt_des, t.float32)
        sm = self._empty((s.shape[0],), t.int32)
        tm = self._empty((s.shape[0],), t.int32)
        cnt = self._empty((1,), t.int32)
        _chk(self.lib.bx_mutual(self.handle, self._stream(), self._p(s), C.c_int32(s.shape[0]), self._p(g),
                                C.c_int32(g.shape[0]), self._p(sm), self._p(tm), self._p(cnt)), "bx_mutual")
        return sm, tm, cnt

    def pose_net(self, s_equi, t_equi, s_mids, t_mids, m_dev, max_m, want_logits=False):
        t = self.torch
        s_equi, t_equi = self._dev(s_equi, t.float32), self._dev(t_equi, t.float32)
        s_mids, t_mids, m_dev = self._dev(s_mids, t.int32), self._dev(t_mids, t.int32), self._dev(m_dev, t.int32)
        ind = self._empty((max_m,), t.float32)
        logits = self._empty((max_m, 2, 1, 16), t.float32) if want_logits else None
        _chk(self.lib.bx_pose_net(self.handle, self._stream(), self._p(s_equi), self._p(t_equi), self._p(s_mids), self._p(t_mids),
                                  self._p(m_dev), C.c_int32(max_m), self._p(ind), self._p(logits)), "bx_pose_net")
        return ind, logits

    def hypotheses(self, ind, s_mids, t_mids, m_dev, max_m, s_R, t_R, s_k, t_k):
        t = self.torch
        a = [self._dev(ind, t.float32), self._dev(s_mids, t.int32), self._dev(t_mids, t.int32), self._dev(m_dev, t.int32)]
        b = [self._dev(v, t.float32) for v in (s_R, t_R, s_k, t_k)]
        R = self._empty((max_m, 9), t.float32)
        tt_ = self._empty((max_m, 3), t.float32)
        ss = self._empty((max_m, 3), t.float32)
        tg = self._empty((max_m, 3), t.float32)
        _chk(self.lib.bx_hypotheses(self.handle, self._stream(), self._p(a[0]), self._p(a[1]), self._p(a[2]), self._p(a[3]),
                                    C.c_int32(max_m), self._p(b[0]), self._p(b[1]), self._p(b[2]), self._p(b[3]), self._p(R),
                                    self._p(tt_), self._p(ss), self._p(tg)), "bx_hypotheses")
        return R, tt_, ss, tg

    def consensus(self, R, tr, ss, tt, M_dev, max_M):
        t = self.torch
        R, tr, ss, tt = [self._dev(v, t.float32) for v in (R, tr, ss, tt)]
        M_dev = self._dev(M_dev, t.int32)
        inl = self._empty((max(max_M, 1),), t.int32)
        cnt = self._empty((1,), t.int32)
        best = self._empty((1,), t.int32)
        _chk(self.lib.bx_consensus(self.handle, self._stream(), self._p(R), self._p(tr), self._p(ss), self._p(tt), self._p(M_dev),
                                   C.c_int32(max_M), self._p(inl), self._p(cnt), self._p(best)), "bx_consensus")
        return inl, cnt, best

    def ransac(self, ss, tt, corr, C_dev, max_C, seed):
        t = self.torch
        ss, tt = self._dev(ss, t.float32), self._dev(tt, t.float32)
        corr, C_dev = self._dev(corr, t.int32), self._dev(C_dev, t.int32)
        T = self._empty((16,), t.float64)
        info = self._empty((2,), t.int32)
        _chk(self.lib.bx_ransac(self.handle, self._stream(), self._p(ss), self._p(tt), self._p(corr), self._p(C_dev),
                                C.c_int32(max_C), C.c_uint64(seed & (2**64 - 1)), self._p(T), self._p(info)), "bx_ransac")
        return T, info

    def kiss_solve(self, ss, tt, corr, C_dev, max_C):
        """KISS-Matcher back-end on the correspondences corr (context created with cfg.match.pose_estimator == "kiss_matcher")"""
        t = self.torch
        ss, tt = self._dev(ss, t.float32), self._dev(tt, t.float32)
        corr, C_dev = self._dev(corr, t.int32), self._dev(C_dev, t.int32)
        T = self._empty((16,), t.float64)
        info = self._empty((4,), t.int32)
        _chk(self.lib.bx_kiss_solve(self.handle, self._stream(), self._p(ss), self._p(tt), self._p(corr), self._p(C_dev),
                                    C.c_int32(max_C), self._p(T), self._p(info)), "bx_kiss_solve")
        return T, info

    def refine(self, ss, tt, M_dev, max_M, T):
        t = self.torch
        ss, tt = self._dev(ss, t.float32), self._dev(tt, t.float32)
        M_dev = self._dev(M_dev, t.int32)
        T = self._dev(T, t.float32).reshape(16).clone()
        it = self._empty((1,), t.int32)
        _chk(self.lib.bx_refine(self.handle, self._stream(), self._p(ss), self._p(tt), self._p(M_dev), C.c_int32(max_M),
                                self._p(T), self._p(it)), "bx_refine")
        return T, it

    def set_capture(self, scale, cloud, n_points):
        """Allocate capture buffers for (scale, cloud) and arm bx_set_capture; returns {name: device tensor}.
        n_points = size of the captured cloud.  set_capture(None, ...) disarms."""
        t = self.torch
        if scale is None:
            _chk(self.lib.bx_set_capture(self.handle, None), "bx_set_capture")
            self._cap = None
            return None
        K, P, S = self.params.num_fps, self.params.num_points_per_patch, self.params.num_scales
        f32, i32 = t.float32, t.int32
        z = lambda shape, dt: t.zeros(shape, dtype=dt, device=f"cuda:{self.device}")
        buf = dict(pts_perm=z((n_points, 3), f32), patches=z((K, P, 3), f32), feat=z((K, 3, 140, 16), f32),
                   x=z((K, 2, 140, 16), f32), kpts=[z((K, 3), f32) for _ in range(2)], desc=[z((K, 32), f32) for _ in range(2)],
                   equi=[z((K, 140, 32), f32) for _ in range(2)], R=[z((K, 9), f32) for _ in range(2)],
                   s_mids=z((K,), i32), t_mids=z((K,), i32), ind=z((K,), f32), R_cat=z((S * K, 9), f32),
                   t_cat=z((S * K, 3), f32), ss_cat=z((S * K, 3), f32), tt_cat=z((S * K, 3), f32), cons_cnt=z((S * K,), i32),
                   inlier_ind=z((S * K,), i32), counts=z((4,), i32), T_ransac=z((16,), t.float64))
        cap = BxCapture()
        cap.scale, cap.cloud = int(scale), int(cloud)
        for k, v in buf.items():
            if isinstance(v, list):
                arr = getattr(cap, k)
                for j, tv in enumerate(v):
                    arr[j] = tv.data_ptr()
            else:
                setattr(cap, k, v.data_ptr())
        _chk(self.lib.bx_set_capture(self.handle, C.byref(cap)), "bx_set_capture")
        self._cap = buf
        return buf

    def new_result(self):
        """BxResult living in pinned host memory (so the final D2H copy is truly asynchronous)."""
        buf = self.torch.zeros(C.sizeof(BxResult), dtype=self.torch.uint8).pin_memory()
        res = BxResult.from_address(buf.data_ptr())
        res._pinned = buf
        return res

    # ---------------------------------------------------------------- whole pair (async)
    def register_pair_async(self, src, tgt, aligned_z, perm_src, perm_tgt, seed, result=None):
        """Enqueue one pair on the current stream.  Returns the BxResult (valid after stream sync)."""
        t = self.torch
        src, tgt = self._dev(src, t.float32), self._dev(tgt, t.float32)
        perm_src, perm_tgt = self._dev(perm_src, t.int32), self._dev(perm_tgt, t.int32)
        res = result if result is not None else self.new_result()
        _chk(self.lib.bx_register_pair(self.handle, self._stream(), self._p(src), C.c_int32(src.shape[0]), self._p(tgt),
                                       C.c_int32(tgt.shape[0]), C.c_int32(int(aligned_z)), self._p(perm_src), self._p(perm_tgt),
                                       C.c_uint64(seed & (2**64 - 1)), C.byref(res)), "bx_register_pair")
        self._keep = [src, tgt, perm_src, perm_tgt, res]
        return res

    # the same in two calls, the early-exit decision taken on the host (include/bufferx.h: bx_register_pair_begin / _finish)
    def new_exit_flag(self):
        """pinned int32[1] the first call's asynchronous copy lands in (1 = the pair left at scale 0)"""
        return self.torch.zeros(1, dtype=self.torch.int32).pin_memory()

    def register_pair_begin_async(self, src, tgt, aligned_z, perm_src, perm_tgt, seed, flag):
        """Enqueue a pair up to the exit test of scale 0 (every scale when no early exit can follow) on the current stream; `flag`
        (new_exit_flag()) holds the decision once the stream has passed this call.  Follow with register_pair_finish_async."""
        t = self.torch
        src, tgt = self._dev(src, t.float32), self._dev(tgt, t.float32)
        perm_src, perm_tgt = self._dev(perm_src, t.int32), self._dev(perm_tgt, t.int32)
        _chk(self.lib.bx_register_pair_begin(self.handle, self._stream(), self._p(src), C.c_int32(src.shape[0]), self._p(tgt),
                                             C.c_int32(tgt.shape[0]), C.c_int32(int(aligned_z)), self._p(perm_src), self._p(perm_tgt),
                                             C.c_uint64(seed & (2**64 - 1)), C.c_void_p(flag.data_ptr())), "bx_register_pair_begin")
        self._keep = [src, tgt, perm_src, perm_tgt, flag]
        return flag

    def register_pair_finish_async(self, exited, result=None):
        """Enqueue the rest of the pending pair: the later scales unless `exited`, the final pose estimation, the refinement."""
        res = result if result is not None else self.new_result()
        _chk(self.lib.bx_register_pair_finish(self.handle, self._stream(), C.c_int32(int(bool(exited))), C.byref(res)), "bx_register_pair_finish")
        self._keep = list(getattr(self, "_keep", [])) + [res]
        return res

    def register_pair_two_calls(self, src, tgt, aligned_z, perm_src, perm_tgt, seed):
        """begin -> synchronise -> finish -> synchronise (the reference's own control flow, models/BUFFERX.py:424-457)"""
        flag = self.register_pair_begin_async(src, tgt, aligned_z, perm_src, perm_tgt, seed, self.new_exit_flag())
        st = self.torch.cuda.current_stream(self.device)
        st.synchronize()
        res = self.register_pair_finish_async(int(flag[0]))
        st.synchronize()
        if res.status != 0:
            raise BxError(f"device-side failure bits 0x{res.status:x}")
        return res

    def register_pair(self, src, tgt, aligned_z, perm_src, perm_tgt, seed):
        res = self.register_pair_async(src, tgt, aligned_z, perm_src, perm_tgt, seed)
        self.torch.cuda.current_stream(self.device).synchronize()
        if res.status != 0:
            raise BxError(f"device-side failure bits 0x{res.status:x}")
        return res
