"""Deterministic elementary functions of the oracle (oracle/bxo_detmath.h) vs numpy / LAPACK."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SHIM = r"""
#include "bxo_detmath.h"
void t_exp(const double*x,int n,double*o){for(int i=0;i<n;++i)o[i]=bxo_exp(x[i]);}
void t_sincos(const double*x,int n,double*s,double*c){for(int i=0;i<n;++i)bxo_sincos(x[i],s+i,c+i);}
void t_acos(const double*x,int n,double*o){for(int i=0;i<n;++i)o[i]=bxo_acos(x[i]);}
void t_log(const double*x,int n,double*o){for(int i=0;i<n;++i)o[i]=bxo_log(x[i]);}
void t_jacobi(const double*a,double*v,double*w){double t[9];for(int i=0;i<9;++i)t[i]=a[i];bxo_jacobi3(t,v,w);}
int t_kabsch(const double*H,double*R){return bxo_kabsch_from_H(H,R);}
"""


def _lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("detmath")
    src = d / "shim.c"
    src.write_text(SHIM)
    so = d / "shim.so"
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"),
                           str(src), "-o", str(so), "-lm"])
    return C.CDLL(str(so))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_detmath(tmp_path_factory):
    L = _lib(tmp_path_factory)
    rng = np.random.default_rng(0)
    x = -rng.random(5000) * 90
    o = np.zeros_like(x)
    L.t_exp(_p(x), len(x), _p(o))
    assert np.allclose(o, np.exp(x), rtol=2e-15, atol=0)
    x = rng.random(5000) * 14 - 7
    s, c = np.zeros_like(x), np.zeros_like(x)
    L.t_sincos(_p(x), len(x), _p(s), _p(c))
    assert np.abs(s - np.sin(x)).max() < 3e-16 and np.abs(c - np.cos(x)).max() < 3e-16
    x = np.concatenate([rng.random(5000) * 2 - 1, [1.0, -1.0, 0.0, 0.5, -0.5, 1 - 1e-12]])
    o = np.zeros_like(x)
    L.t_acos(_p(x), len(x), _p(o))
    assert np.abs(o - np.arccos(x)).max() < 2e-15
    x = np.concatenate([rng.random(5000), [1e-300, 1.0, 0.001, 1 - 1e-9]])
    x = x[x > 0]
    o = np.zeros_like(x)
    L.t_log(_p(x), len(x), _p(o))
    assert np.allclose(o, np.log(x), rtol=4e-15, atol=1e-16)
    # fp32 results agree with correctly-rounded libm results to <= 1 ulp
    xf = (rng.random(2000) * 6.3).astype(np.float32)
    s, c = np.zeros(2000), np.zeros(2000)
    L.t_sincos(_p(xf.astype(np.float64)), 2000, _p(s), _p(c))
    assert np.abs(s.astype(np.float32) - np.sin(xf.astype(np.float64)).astype(np.float32)).max() <= 6e-8


def test_jacobi_and_kabsch(tmp_path_factory):
    L = _lib(tmp_path_factory)
    rng = np.random.default_rng(1)
    for _ in range(200):
        B = rng.standard_normal((3, 3))
        A = B @ B.T
        v, w = np.zeros(9), np.zeros(3)
        L.t_jacobi(_p(np.ascontiguousarray(A)), _p(v), _p(w))
        V = v.reshape(3, 3)
        assert np.allclose(V @ np.diag(w) @ V.T, A, atol=1e-12)
        assert np.allclose(np.sort(w), np.linalg.eigvalsh(A), atol=1e-12)
    for rank2 in (False, True):
        for _ in range(100):
            n = 3 if rank2 else 20
            a = rng.standard_normal((n, 3))
            Rg, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            if np.linalg.det(Rg) < 0:
                Rg[:, 0] *= -1
            b = a @ Rg.T
            a0, b0 = a - a.mean(0), b - b.mean(0)
            H = np.ascontiguousarray(a0.T @ b0)
            R = np.zeros(9)
            assert L.t_kabsch(_p(H), _p(R)) == 1
            assert np.allclose(R.reshape(3, 3), Rg, atol=1e-9)
    # reflection case: the closest PROPER rotation is returned (det = +1)
    a = rng.standard_normal((30, 3))
    b = a * np.array([1, 1, -1.0])
    R = np.zeros(9)
    L.t_kabsch(_p(np.ascontiguousarray(a.T @ b)), _p(R))
    assert abs(np.linalg.det(R.reshape(3, 3)) - 1) < 1e-9
    U, S, Vt = np.linalg.svd(a.T @ b)
    Rref = Vt.T @ np.diag([1, 1, np.linalg.det(Vt.T @ U.T)]) @ U.T
    assert np.allclose(R.reshape(3, 3), Rref, atol=1e-9)
    # rank < 2 is rejected
    assert L.t_kabsch(_p(np.zeros(9)), _p(R)) == 0


# ------------------------------------------------------------------ KISS-Matcher back-end (restated; oracle/bx_oracle.c bxo_kiss_solve)
def _kiss_case(seed, M, outlier_frac, noise, structured=0.0):
    rng = np.random.default_rng(seed)
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = rng.uniform(0.2, 1.2)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rg = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    tg = rng.uniform(-1, 1, 3)
    s = (rng.random((M, 3)) * 6).astype(np.float32)
    g = (s @ Rg.T + tg + noise * rng.standard_normal((M, 3))).astype(np.float32)
    bad = rng.random(M) < outlier_frac
    g[bad] = (rng.random((int(bad.sum()), 3)) * 6).astype(np.float32)
    if structured > 0:                       # a second, smaller consistent motion: must lose against the larger one
        k = int(M * structured)
        R2 = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
        g[:k] = (s[:k] @ R2.T + np.array([3.0, 0.5, -1.0])).astype(np.float32)
        bad[:k] = True
    return s, g, Rg, tg, bad


def test_kiss_solve_recovers_pose_under_outliers():
    from oracle import oracle as O
    for seed, M, frac, structured in ((0, 500, 0.7, 0.0), (1, 300, 0.5, 0.15), (2, 800, 0.8, 0.0)):
        s, g, Rg, tg, bad = _kiss_case(seed, M, frac, 0.02, structured)
        T, info = O.kiss_solve(s, g, np.arange(M, dtype=np.int32), 0.3)
        good = int((~bad).sum())
        assert info[1] >= 0.9 * good and info[0] >= 0.8 * good, (info, good)
        assert np.abs(T[:3, :3] - Rg).max() < 5e-3 and np.abs(T[:3, 3] - tg).max() < 2e-2
        assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-9


def test_kiss_solve_gnc_iterates_and_degenerate_inputs():
    from oracle import oracle as O
    # noise well above the bound for a third of the "inliers": the core keeps some of them, GNC has to down-weight them
    s, g, Rg, tg, bad = _kiss_case(5, 400, 0.3, 0.02)
    rng = np.random.default_rng(9)
    loose = np.flatnonzero(~bad)[::3]
    g[loose] += (0.35 * rng.standard_normal((len(loose), 3))).astype(np.float32)
    T, info = O.kiss_solve(s, g, np.arange(400, dtype=np.int32), 0.3)
    assert info[3] > 1 and info[2] <= info[1] and info[0] <= info[2]
    assert np.abs(T[:3, :3] - Rg).max() < 2e-2
    # fewer than two correspondences / a single pair / collinear sources: identity or a finite pose, never a crash
    for C in (0, 1):
        T, info = O.kiss_solve(s, g, np.arange(C, dtype=np.int32), 0.3)
        assert np.array_equal(T, np.eye(4)) and info[0] == 0
    line = np.stack([np.linspace(0, 5, 50), np.zeros(50), np.zeros(50)], 1).astype(np.float32)
    T, info = O.kiss_solve(line, line + np.float32(1.0), np.arange(50, dtype=np.int32), 0.3)
    assert np.isfinite(T).all()


# ------------------------------------------------------------------ round 3: the two re-associated stages against their direct forms
def test_winograd_layers_close_to_direct_form(oracle, bx, packed):
    """bxo_conv_wino (F(2x2, 3x3), the contract of k_wino.hip) against bxo_conv on the cylindrical tap table, every layer of
    Cylindrical_Net on post-ReLU-like inputs: same layer up to the re-association (<= 2e-5 absolute on values of O(5)); elevation
    padding, azimuth wrap-around and the dropped 8th output row are covered by comparing ALL 140 positions."""
    rng = np.random.default_rng(0)
    tap = bx.weights.cyl_tap_table()
    x = np.abs(rng.standard_normal((3, 3, 140, 16))).astype(np.float32)
    for L in packed["desc"]:
        a = oracle.conv(x, tap, L["W"], L["b"], L["relu"])
        b = oracle.conv_wino(x, L["W"], L["b"], L["relu"])
        assert a.shape == b.shape and np.abs(a - b).max() < 2e-5 * max(1.0, float(np.abs(a).max()))
        x = a
    # a map that is non-zero in ONE cell: every tile / window position of the transform is hit by some shift of it
    L = packed["desc"][1]
    for p in (0, 19, 20, 79, 120, 139):
        x = np.zeros((1, 4, 140, 16), np.float32)
        x[0, :, p, :] = 1.0
        assert np.abs(oracle.conv(x, tap, L["W"], L["b"], True) - oracle.conv_wino(x, L["W"], L["b"], True)).max() < 1e-5


def test_winograd43_layers_close_to_direct_form(oracle, bx, packed):
    """bxo_conv_wino43 (F(4x4, 3x3), the contract of k_wino43.hip and the default form of the >= 64-channel layers) against bxo_conv:
    the same layer up to re-association and the amplification of the F(4x4) transforms (<= 1e-4 absolute on values of O(5); measured
    ~3e-5: tests/study_wino43_error.py); all 140 positions (elevation padding, wrap-around, the dropped 8th output row)."""
    rng = np.random.default_rng(0)
    tap = bx.weights.cyl_tap_table()
    x = np.abs(rng.standard_normal((4, 3, 140, 16))).astype(np.float32)     # 4 units: one full group of three + a group of one
    for L in packed["desc"]:
        a = oracle.conv(x, tap, L["W"], L["b"], L["relu"])
        b = oracle.conv_wino43(x, L["W"], L["b"], L["relu"])
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-4 * max(1.0, float(np.abs(a).max()) / 5.0)
        x = a
    L = packed["desc"][1]
    for p in (0, 3, 19, 20, 79, 83, 120, 139):
        x = np.zeros((1, 4, 140, 16), np.float32)
        x[0, :, p, :] = 1.0
        assert np.abs(oracle.conv(x, tap, L["W"], L["b"], True) - oracle.conv_wino43(x, L["W"], L["b"], True)).max() < 2e-5


def test_mixed_tile_winograd_layers(oracle, bx, packed):
    """bxo_conv_wino43m (round 6, the contract of k_wino43m.hip): the map rows 0..3 are the F(4x4, 3x3) tiles of bxo_conv_wino43 BIT FOR BIT;
    the rows 4..6 are F(3x4, 3x3) tiles (F(3, 3) on the points {1, -1, 1/2, -1/2, inf}) and agree with the direct form to the same bound
    as the all-F(4x4) form; unit impulses at the positions that touch the seam between the two tile rows, the elevation padding and the
    azimuth wrap-around."""
    rng = np.random.default_rng(0)
    tap = bx.weights.cyl_tap_table()
    x = np.abs(rng.standard_normal((4, 3, 140, 16))).astype(np.float32)
    for L in packed["desc"]:
        a = oracle.conv(x, tap, L["W"], L["b"], L["relu"])
        b = oracle.conv_wino43(x, L["W"], L["b"], L["relu"])
        m = oracle.conv_wino43m(x, L["W"], L["b"], L["relu"])
        assert np.array_equal(m.reshape(4, -1, 7, 20, 16)[:, :, :4], b.reshape(4, -1, 7, 20, 16)[:, :, :4])
        assert a.shape == m.shape and np.abs(a - m).max() < 1e-4 * max(1.0, float(np.abs(a).max()) / 5.0)
        x = a
    L = packed["desc"][1]
    for p in (60, 63, 79, 80, 99, 100, 119, 120, 123, 139):      # rows 3..6: the seam, the last row, both wrap-around columns
        x = np.zeros((1, 4, 140, 16), np.float32)
        x[0, :, p, :] = 1.0
        assert np.abs(oracle.conv(x, tap, L["W"], L["b"], True) - oracle.conv_wino43m(x, L["W"], L["b"], True)).max() < 2e-5


def test_valid_winograd43_layers_close_to_direct_form(oracle, bx, packed):
    """bxo_conv_wino43_valid (valid F(4x4, 3x3), the contract of k_wino43v.hip and the default form of CostNet layers 1..5) against bxo_conv
    and against the F(2x2) form: D = 18 (folded k rows), 16 and 12 (last tiles reach beyond the map: zero rows / dropped outputs), 14, 10;
    chained like the network; and a one-hot map at the corners / the partial last tile of every layer."""
    rng = np.random.default_rng(2)
    geo = bx.weights.pose_geometry()
    dims0 = geo[1][0]
    x = np.abs(rng.standard_normal((5, 2, int(np.prod(dims0)), 16))).astype(np.float32)
    for layer in range(1, 6):
        L = packed["pose"][layer]
        dims, k, _ = geo[layer]
        tap, _ = bx.weights.valid_tap_table(dims, k)
        a = oracle.conv(x, tap, L["W"], L["b"], L["relu"])
        b = oracle.conv_wino43_valid(x, L["W"], L["b"], L["relu"], dims[0], dims[1])
        c = oracle.conv_wino_valid(x, L["W"], L["b"], L["relu"], dims[0], dims[1])
        scale = max(1.0, float(np.abs(a).max()) / 5.0)
        assert a.shape == b.shape == c.shape
        assert np.abs(a - b).max() < 1e-4 * scale and np.abs(a - c).max() < 2e-5 * scale, (layer, np.abs(a - b).max(), np.abs(a - c).max())
        D, fold = dims[0], dims[1]
        for (n, l) in ((0, 0), (0, D - 1), (D - 1, 0), (D - 1, D - 1), (D - 3, D - 3), (4, 3)):
            h = np.zeros((1, L["W"].shape[0], D * fold * D, 16), np.float32)
            for kk in range(fold):
                h[0, :, (n * fold + kk) * D + l, :] = 1.0
            assert np.abs(oracle.conv(h, tap, L["W"], L["b"], True) - oracle.conv_wino43_valid(h, L["W"], L["b"], True, D, fold)).max() < 2e-5, (layer, n, l)
        x = a


def test_collapsed_cost_layer_close_to_direct_form(oracle, bx, packed):
    """bxo_cost_l0 (binary64 P - Q form, the contract of k_cost.hip) against the fp32 convolution of the materialised cost volume
    (CostVolume.forward + the first Conv3d, models/BUFFERX.py:59-65, models/patchnet.py:196), incl. nearly equal maps (P - Q cancels)."""
    rng = np.random.default_rng(1)
    K, m = 12, 9
    se = rng.standard_normal((K, 140, 32)).astype(np.float32)
    te = rng.standard_normal((K, 140, 32)).astype(np.float32)
    te[:4] = se[:4] + 1e-3 * rng.standard_normal((4, 140, 32)).astype(np.float32)
    sm = np.arange(m, dtype=np.int32)
    tm = np.r_[np.arange(4), rng.permutation(K)[:m - 4]].astype(np.int32)
    L0 = packed["pose"][0]
    tap, _ = bx.weights.valid_tap_table((20, 5, 20), (3, 3, 3))
    direct = oracle.conv(oracle.cost_volume(se, te, sm, tm), tap, L0["W"], L0["b"], True)
    coll = oracle.cost_l0(se, te, sm, tm, L0["W"], L0["b"])
    assert direct.shape == coll.shape == (m, 2, 972, 16)
    assert np.abs(direct - coll).max() < 3e-5 * max(1.0, float(np.abs(direct).max()))


# ------------------------------------------------------------------ distance form study switch (round 5)
def test_distance_form_switch(oracle, monkeypatch):
    """The STUDY switch that evaluates the squared distances of FPS / ball_query the way nvcc contracts the upstream kernels
    (fmaf(dz, dz, fmaf(dy, dy, dx * dx))): the C oracle (hardware fmaf) and the numpy stand-ins of tests/golden/ref_harness.py
    (binary64 emulation) must agree with each other in both forms, the switch must restore the contract's form, and the two forms
    must differ only where a distance sits within an ulp of a threshold or of another distance (profiles/r05_distance_form.jsonl has
    the real-size count: 1 neighbour list of 30 000, no FPS index, no match)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_harness as H
    rng = np.random.default_rng(5)
    pts = (rng.random((6000, 3), np.float32) * 3).astype(np.float32)
    kp_idx = {}
    for form in ("unfused", "nvcc_fma"):
        monkeypatch.setattr(H, "DIST_FORM", form)
        with oracle.distance_form(form):
            a = oracle.fps(pts, 600)
            kp = pts[a[:200]]
            i, _ = oracle.ball_group(pts, kp, np.float32(0.31), 128)
        assert np.array_equal(H._fps_np(pts, 600), a), form
        assert np.array_equal(H._ball_query_np(0.31, 128, pts, kp), i), form
        kp_idx[form] = (a, i)
    assert int(oracle.lib().bxo_get_distance_form()) == 0          # the context manager restored the contract's form
    assert (kp_idx["unfused"][0] != kp_idx["nvcc_fma"][0]).sum() <= 4
