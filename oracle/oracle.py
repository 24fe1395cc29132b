"""oracle/oracle.py -- TEST INFRASTRUCTURE: ctypes front-end of the CPU oracle (oracle/bx_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (buffer-x_amd/) never does.

Besides thin wrappers around each C stage, `register_pair()` chains the stages exactly like
BufferX.forward's inference branch (reference models/BUFFERX.py:257-467) so that the HIP library's
`bx_register_pair` can be compared end-to-end.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists). Building != using."""
    so = os.path.join(_HERE, "_build", "libbx_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("bx_oracle.c", "bxo_detmath.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if stale or force:
        subprocess.check_call(["make", "-C", _HERE, "_build/libbx_oracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref_neighbors.so")
    shim = os.path.join(_HERE, "ref_neighbors_shim.cpp")
    ref_stale = (not os.path.exists(ref_so)) or os.path.getmtime(shim) > os.path.getmtime(ref_so)
    if os.path.isdir("/root/reference/cpp_wrappers") and (force or ref_stale):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.bxo_radius.restype = C.c_double
        _LIB.bxo_mix64.restype = C.c_uint64
        _LIB.bxo_mix64.argtypes = [C.c_uint64, C.c_uint64]
    return _LIB


def ref_lib():
    """The reference's own nanoflann radius search (compiled from /root/reference). None if absent."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_neighbors.so")
        if not os.path.exists(p):
            try:
                build()
            except Exception:
                pass
        if os.path.exists(p):
            _REF = C.CDLL(p)
    return _REF


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------ stage wrappers
def mix64(seed, ctr):
    return int(lib().bxo_mix64(C.c_uint64(seed & (2**64 - 1)), C.c_uint64(ctr)))


def make_perm(n, seed, stream):
    """Deterministic permutation standing in for np.random.choice(N, N, replace=False)
    (reference models/patch_embedder.py:96): argsort of counter-RNG keys."""
    keys = np.array([mix64(seed, (stream << 32) + i) for i in range(n)], dtype=np.uint64)
    return np.argsort(keys, kind="stable").astype(np.int32)


class distance_form:
    """`with oracle.distance_form("nvcc_fma"): ...` -- the STUDY switch of bx_oracle.c (squared distances of FPS / ball query / SPT as
    nvcc's default contraction would evaluate them); the contract, and the product, are "unfused"."""
    FORMS = ("unfused", "nvcc_fma")

    def __init__(self, name):
        self.new = self.FORMS.index(name)

    def __enter__(self):
        L = lib()
        self.old = int(L.bxo_get_distance_form())
        L.bxo_set_distance_form(C.c_int(self.new))
        return self

    def __exit__(self, *a):
        lib().bxo_set_distance_form(C.c_int(self.old))


def fps(xyz, m):
    xyz = _f(xyz)
    idx = np.zeros(m, np.int32)
    lib().bxo_fps(_p(xyz), C.c_int(len(xyz)), C.c_int(m), _p(idx))
    return idx


def radius(pts, n_orig, kpts, threshold):
    pts, kpts = _f(pts), _f(kpts)
    return float(lib().bxo_radius(_p(pts), C.c_int(len(pts)), C.c_long(n_orig), _p(kpts), C.c_int(len(kpts)),
                                  C.c_double(threshold)))


def ball_group(pts_perm, kpts, radius_, P):
    pts_perm, kpts = _f(pts_perm), _f(kpts)
    K = len(kpts)
    idx = np.zeros((K, P), np.int32)
    patches = np.zeros((K, P, 3), np.float32)
    lib().bxo_ball_group(_p(pts_perm), C.c_int(len(pts_perm)), _p(kpts), C.c_int(K), C.c_float(radius_), C.c_int(P),
                         _p(idx), _p(patches))
    return idx, patches


def voxel_table(rad_n=3, ele_n=7, azi_n=20):
    cen = np.zeros((rad_n * ele_n * azi_n, 3), np.float32)
    rot = np.zeros((azi_n, 4), np.float32)
    lib().bxo_voxel_table(C.c_int(rad_n), C.c_int(ele_n), C.c_int(azi_n), _p(cen), _p(rot))
    return cen, rot


def patch_features(patches, des_r, aligned, pnt_w, pnt_b, rad_n=3, ele_n=7, azi_n=20, nsample=10, delta=0.8,
                   debug=False):
    patches = _f(patches)
    K, P, _ = patches.shape
    cen, rot = voxel_table(rad_n, ele_n, azi_n)
    V = rad_n * ele_n * azi_n
    R = np.zeros((K, 9), np.float32)
    feat = np.zeros((K, rad_n, ele_n * azi_n, 16), np.float32)
    dn = np.zeros((K, P, 3), np.float32) if debug else None
    ds = np.zeros((K, V, nsample, 3), np.float32) if debug else None
    pw, pb = _f(pnt_w), _f(pnt_b)
    voxel_r = np.float32(delta / rad_n)
    lib().bxo_patch_features(_p(patches), C.c_int(K), C.c_int(P), C.c_float(np.float32(des_r)), C.c_int(int(aligned)),
                             _p(cen), _p(rot), C.c_int(rad_n), C.c_int(ele_n), C.c_int(azi_n), C.c_int(nsample),
                             C.c_float(voxel_r), _p(pw), _p(pb), _p(R), _p(feat), _p(dn), _p(ds))
    if debug:
        return R, feat, dn, ds
    return R, feat


def conv(x, tap, W, bias, relu):
    """x [units][n_chunks][p_in][16]; tap [ntaps][p_out] int32; W [n_chunks][ntaps][16][cout]."""
    x, W, bias, tap = _f(x), _f(W), _f(bias), _i(tap)
    units, n_chunks, p_in, _ = x.shape
    ntaps, p_out = tap.shape
    cout = W.shape[-1]
    assert W.shape == (n_chunks, ntaps, 16, cout), (W.shape, (n_chunks, ntaps, 16, cout))
    out = np.zeros((units, (cout + 15) // 16, p_out, 16), np.float32)
    lib().bxo_conv(_p(x), C.c_int(units), C.c_int(n_chunks), C.c_int(p_in), _p(tap), C.c_int(ntaps), C.c_int(p_out),
                   _p(W), _p(bias), C.c_int(cout), C.c_int(int(relu)), _p(out))
    return out


def conv_wino(x, W, bias, relu, ele_n=7, azi_n=20):
    """Cylindrical 3x3 layer as Winograd F(2x2, 3x3) (bxo_conv_wino).  x [units][n_chunks][ele*azi][16]; W [n_chunks][9][16][cout]."""
    x, W, bias = _f(x), _f(W), _f(bias)
    units, n_chunks, p_in, _ = x.shape
    cout = W.shape[-1]
    assert p_in == ele_n * azi_n and W.shape == (n_chunks, 9, 16, cout)
    out = np.zeros((units, (cout + 15) // 16, p_in, 16), np.float32)
    lib().bxo_conv_wino(_p(x), C.c_int(units), C.c_int(n_chunks), C.c_int(ele_n), C.c_int(azi_n), _p(W), _p(bias), C.c_int(cout),
                        C.c_int(int(relu)), _p(out))
    return out


def conv_wino43(x, W, bias, relu, ele_n=7, azi_n=20):
    """Cylindrical 3x3 layer as Winograd F(4x4, 3x3) (bxo_conv_wino43).  x [units][n_chunks][ele*azi][16]; W [n_chunks][9][16][cout]."""
    x, W, bias = _f(x), _f(W), _f(bias)
    units, n_chunks, p_in, _ = x.shape
    cout = W.shape[-1]
    assert p_in == ele_n * azi_n and azi_n % 4 == 0 and W.shape == (n_chunks, 9, 16, cout)
    out = np.zeros((units, (cout + 15) // 16, p_in, 16), np.float32)
    lib().bxo_conv_wino43(_p(x), C.c_int(units), C.c_int(n_chunks), C.c_int(ele_n), C.c_int(azi_n), _p(W), _p(bias), C.c_int(cout),
                          C.c_int(int(relu)), _p(out))
    return out


def conv_wino43m(x, W, bias, relu, ele_n=7, azi_n=20):
    """Cylindrical 3x3 layer in the MIXED-tile Winograd form (bxo_conv_wino43m): F(4x4, 3x3) on the output rows 0..3 (bit-identical to
    conv_wino43 there), F(3x4, 3x3) on the rows 4..6.  Arguments as conv_wino43."""
    x, W, bias = _f(x), _f(W), _f(bias)
    units, n_chunks, p_in, _ = x.shape
    cout = W.shape[-1]
    assert p_in == ele_n * azi_n and azi_n % 4 == 0 and ele_n == 7 and W.shape == (n_chunks, 9, 16, cout)
    out = np.zeros((units, (cout + 15) // 16, p_in, 16), np.float32)
    lib().bxo_conv_wino43m(_p(x), C.c_int(units), C.c_int(n_chunks), C.c_int(ele_n), C.c_int(azi_n), _p(W), _p(bias), C.c_int(cout),
                           C.c_int(int(relu)), _p(out))
    return out


def conv_wino_valid(x, W, bias, relu, D, fold):
    """CostNet layer as a valid Winograd F(2x2, 3x3) convolution over a D x D map (bxo_conv_wino_valid); fold = 3 for the k(3,3,3)
    layer (its three k rows become input channels), 1 for the k(3,1,3) layers.  x [units][n_chunks][D*fold*D][16]."""
    x, W, bias = _f(x), _f(W), _f(bias)
    units, n_chunks, p_in, _ = x.shape
    cout = W.shape[-1]
    assert p_in == D * fold * D and W.shape == (n_chunks, 9 * fold, 16, cout), (x.shape, W.shape, D, fold)
    out = np.zeros((units, (cout + 15) // 16, (D - 2) * (D - 2), 16), np.float32)
    lib().bxo_conv_wino_valid(_p(x), C.c_int(units), C.c_int(n_chunks), C.c_int(D), C.c_int(fold), _p(W), _p(bias), C.c_int(cout),
                              C.c_int(int(relu)), _p(out))
    return out


def conv_wino43_valid(x, W, bias, relu, D, fold):
    """CostNet layer as a valid Winograd F(4x4, 3x3) convolution over a D x D map (bxo_conv_wino43_valid); arguments as conv_wino_valid."""
    x, W, bias = _f(x), _f(W), _f(bias)
    units, n_chunks, p_in, _ = x.shape
    cout = W.shape[-1]
    assert p_in == D * fold * D and W.shape == (n_chunks, 9 * fold, 16, cout), (x.shape, W.shape, D, fold)
    out = np.zeros((units, (cout + 15) // 16, (D - 2) * (D - 2), 16), np.float32)
    lib().bxo_conv_wino43_valid(_p(x), C.c_int(units), C.c_int(n_chunks), C.c_int(D), C.c_int(fold), _p(W), _p(bias), C.c_int(cout),
                                C.c_int(int(relu)), _p(out))
    return out


def _default_form(key):
    import bufferx_amd.config as _c       # the product's knob table: the default of every arithmetic form lives in ONE place
    return _c.ARITH_DEFAULT[key]


def pose_conv(layer, x, tap, dims, W, bias, relu, form=None):
    """CostNet layer `layer` (1..9) in the arithmetic form `form` of bx_params.pose_conv_form ("winograd43": layers 1..5 as valid
    F(4x4, 3x3) convolutions, wino43v_kernel | "winograd22": valid F(2x2, 3x3), wino_pose_kernel | "direct": conv_kernel); dims = input dims (n, k, l) of the layer."""
    form = form or _default_form("pose_conv")
    assert form in ("winograd43", "winograd22", "direct"), form
    if form == "winograd43" and 1 <= layer <= 5:
        return conv_wino43_valid(x, W, bias, relu, dims[0], dims[1])
    if form == "winograd22" and 1 <= layer <= 5:
        return conv_wino_valid(x, W, bias, relu, dims[0], dims[1])
    return conv(x, tap, W, bias, relu)


def desc_conv(x, tap, W, bias, relu, form=None):
    """One Cylindrical_Net layer in the arithmetic form `form` of bx_params.desc_conv_form: "winograd43" = bxo_conv_wino43 for every
    layer (k_wino43.hip), "winograd43m" = bxo_conv_wino43m (mixed F(4x4) / F(3x4) tiles, k_wino43m.hip), "winograd22" = bxo_conv_wino for the layers with >= 64 output channels (k_wino.hip; the two 32-channel layers
    stay direct), "direct" = fmaf chain over chunk > tap > channel (conv_kernel, k_conv.hip)."""
    form = form or _default_form("desc_conv")
    assert form in ("winograd43", "winograd22", "direct", "winograd43m"), form
    if form == "winograd43":
        return conv_wino43(x, W, bias, relu)
    if form == "winograd43m":      # mixed tiles: F(4x4) rows 0..3, F(3x4) rows 4..6 (k_wino43m.hip), every layer
        return conv_wino43m(x, W, bias, relu)
    if form == "winograd22" and np.asarray(W).shape[-1] >= 64:
        return conv_wino(x, W, bias, relu)
    return conv(x, tap, W, bias, relu)


def desc_head(x, w1, b1, w2, b2):
    x = _f(x)
    K, _, npos, _ = x.shape
    desc = np.zeros((K, 32), np.float32)
    equi = np.zeros((K, npos, 32), np.float32)
    w1, b1, w2, b2 = _f(w1), _f(b1), _f(w2), _f(b2)
    lib().bxo_desc_head(_p(x), C.c_int(K), C.c_int(npos), _p(w1), _p(b1), _p(w2), _p(b2), _p(desc), _p(equi))
    return desc, equi


def mutual(src_des, tgt_des):
    src_des, tgt_des = _f(src_des), _f(tgt_des)
    ns, nt = len(src_des), len(tgt_des)
    s = np.zeros(ns, np.int32)
    t = np.zeros(ns, np.int32)
    snn = np.zeros(ns, np.int32)
    tnn = np.zeros(nt, np.int32)
    m = lib().bxo_mutual(_p(src_des), C.c_int(ns), _p(tgt_des), C.c_int(nt), C.c_int(src_des.shape[1]), _p(s), _p(t),
                         _p(snn), _p(tnn))
    return s[:m].copy(), t[:m].copy(), snn, tnn


def cost_volume(s_equi, t_equi, s_mids, t_mids, ele_n=7, azi_n=20):
    s_equi, t_equi, s_mids, t_mids = _f(s_equi), _f(t_equi), _i(s_mids), _i(t_mids)
    m = len(s_mids)
    out = np.zeros((m, 2, azi_n * (ele_n - 2) * azi_n, 16), np.float32)
    lib().bxo_cost_volume(_p(s_equi), _p(t_equi), _p(s_mids), _p(t_mids), C.c_int(m), C.c_int(ele_n), C.c_int(azi_n),
                          _p(out))
    return out


def cost_l0(s_equi, t_equi, s_mids, t_mids, W0, b0, ele_n=7, azi_n=20):
    """CostNet layer 0 on the implicit cost volume (bxo_cost_l0): [m][2][(A-2)(H-2)(A-2)][16], ReLU applied."""
    s_equi, t_equi, s_mids, t_mids, W0, b0 = _f(s_equi), _f(t_equi), _i(s_mids), _i(t_mids), _f(W0), _f(b0)
    m = len(s_mids)
    out = np.zeros((m, 2, (azi_n - 2) * (ele_n - 4) * (azi_n - 2), 16), np.float32)
    lib().bxo_cost_l0(_p(s_equi), _p(t_equi), _p(s_mids), _p(t_mids), C.c_int(m), C.c_int(ele_n), C.c_int(azi_n), _p(W0), _p(b0),
                      _p(out))
    return out


def soft_argmax(logits, azi_n=20):
    logits = _f(logits)
    m = logits.shape[0]
    ind = np.zeros(m, np.float32)
    lib().bxo_soft_argmax(_p(logits), C.c_int(m), C.c_int(azi_n), _p(ind))
    return ind


def hypotheses(ind, ss_R, tt_R, ss_kpts, tt_kpts, azi_n=20):
    ind, ss_R, tt_R, ss_kpts, tt_kpts = _f(ind), _f(ss_R), _f(tt_R), _f(ss_kpts), _f(tt_kpts)
    m = len(ind)
    R = np.zeros((m, 9), np.float32)
    t = np.zeros((m, 3), np.float32)
    lib().bxo_hypotheses(_p(ind), C.c_int(m), C.c_int(azi_n), _p(ss_R), _p(tt_R), _p(ss_kpts), _p(tt_kpts), _p(R), _p(t))
    return R, t


def consensus(R, t, ss, tt, inlier_th, azi_n=20):
    R, t, ss, tt = _f(R), _f(t), _f(ss), _f(tt)
    M = len(ss)
    ind = np.zeros(max(M, 1), np.int32)
    best = C.c_int32(-1)
    counts = np.zeros(max(M, 1), np.int32)
    c = lib().bxo_consensus(_p(R), _p(t), _p(ss), _p(tt), C.c_int(M), C.c_int(azi_n), C.c_float(np.float32(inlier_th)),
                            _p(ind), C.byref(best), _p(counts))
    return ind[:c].copy(), int(best.value), counts[:M]


def ransac(ss, tt, corr, dist_th, similar_th, confidence, max_iter, seed):
    ss, tt, corr = _f(ss), _f(tt), _i(corr)
    T = np.zeros((4, 4), np.float64)
    it = C.c_int32(0)
    lib().bxo_ransac.restype = C.c_int
    n = lib().bxo_ransac(_p(ss), _p(tt), _p(corr), C.c_int(len(corr)), C.c_double(dist_th), C.c_double(similar_th),
                         C.c_double(confidence), C.c_int(max_iter), C.c_uint64(seed & (2**64 - 1)), _p(T), C.byref(it))
    return T, int(n), int(it.value)


def refine(ss, tt, dist_th, T):
    ss, tt = _f(ss), _f(tt)
    T = np.ascontiguousarray(T, dtype=np.float32).copy()
    it = C.c_int32(0)
    lib().bxo_refine(_p(ss), _p(tt), C.c_int(len(ss)), C.c_float(np.float32(dist_th)), _p(T), C.byref(it))
    return T, int(it.value)


def radius_counts(queries, supports, radius_):
    q, s = _f(queries), _f(supports)
    cnt = np.zeros(len(q), np.int32)
    lib().bxo_radius_count(_p(q), C.c_int(len(q)), _p(s), C.c_int(len(s)), C.c_float(radius_), _p(cnt))
    return cnt


def ref_radius_neighbors(queries, supports, radius_, max_out):
    r = ref_lib()
    if r is None:
        return None
    q, s = _f(queries), _f(supports)
    out = np.zeros((len(q), max_out), np.int32)
    cnt = np.zeros(len(q), np.int32)
    r.ref_radius_neighbors(_p(q), C.c_int(len(q)), _p(s), C.c_int(len(s)), C.c_float(radius_), _p(out), C.c_int(max_out),
                           _p(cnt))
    return out, cnt


def ref_batch_neighbors(queries, supports, radius_):
    """Reference call sequence of batch_nanoflann_neighbors (cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332) for one
    cloud: KD-tree build + serial sorted radius search of every query + the dense padded index matrix.  -> (max_count, checksum)
    or None when oracle/_ref is absent.  bench.py times this call (single thread, like the reference's float path)."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_batch_neighbors"):
        return None
    q, s = _f(queries), _f(supports)
    cs = C.c_longlong(0)
    mc = r.ref_batch_neighbors(_p(q), C.c_int(len(q)), _p(s), C.c_int(len(s)), C.c_float(radius_), C.byref(cs))
    return int(mc), int(cs.value)


def kiss_solve(ss, tt, corr, resolution, robin_gain=1.0, solver_gain=0.75):
    """KISS-Matcher solve() on the correspondences corr (restated from the published algorithm, see bx_oracle.c; parity unpinned
    against the `kiss_matcher` package, which is not under /root/reference).  -> (T 4x4 float64, info[4] = final inliers, core size,
    rotation inliers, GNC iterations)."""
    ss, tt, corr = _f(ss), _f(tt), _i(corr)
    T = np.zeros((4, 4), np.float64)
    info = np.zeros(4, np.int32)
    lib().bxo_kiss_solve.restype = C.c_int
    lib().bxo_kiss_solve(_p(ss), _p(tt), _p(corr), C.c_int(len(corr)), C.c_double(robin_gain * resolution),
                         C.c_double(solver_gain * resolution), _p(T), _p(info))
    return T, info
