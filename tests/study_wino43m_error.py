"""Numerical study (CPU, not a test), round 6: the MIXED-tile Winograd form of the Cylindrical_Net layers -- F(4x4, 3x3) on the output rows
0..3 of the 7 x 20 map and F(3x4, 3x3) on the rows 4..6 (5 x 6 = 30 planes instead of 36: the 8th output row, which the all-F(4x4)
form computes and throws away, is never multiplied) -- against the all-F(4x4) form, the F(2x2) form and the direct fp32 form, error
measured against a binary64 convolution on the same realistic activations as tests/study_wino43_error.py (layer 3, 128 -> 128).
F(3, 3) on the points {0, 1, -1, 2, inf}.  Run:  python tests/study_wino43m_error.py"""
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from study_wino43_error import BT4, G4, AT4, BT2, G2, AT2, f32, wino  # noqa: E402

BT3 = np.array([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]], np.float64)
G3 = np.array([[1 / 2, 0, 0], [-1 / 2, -1 / 2, -1 / 2], [-1 / 6, 1 / 6, -1 / 6], [1 / 6, 1 / 3, 2 / 3], [0, 0, 1]], np.float64)
AT3 = np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]], np.float64)


def wino_rows(x, w, r0, mr, btr, gr, atr):
    """output rows r0 .. r0 + mr - 1 of the cylindrical map with the (btr, gr, atr) transform down the rows and F(4, 3) along the
    (circular) columns; fp32 emulation as study_wino43_error.wino"""
    units, H, W, C = x.shape
    O = w.shape[-1]
    ar, ac, tw = mr + 2, 6, W // 4
    xp = np.zeros((units, H + 4, W + 2, C), np.float32)          # rows -1 .. H + 2
    xp[:, 1:H + 1, 1:W + 1] = x
    xp[:, 1:H + 1, 0] = x[:, :, W - 1]
    xp[:, 1:H + 1, W + 1] = x[:, :, 0]
    U = f32(np.einsum("ik,klco,jl->ijco", gr, w, G4)).reshape(ar * ac, C, O)
    d = np.empty((units, tw, ar, ac, C), np.float64)
    for c in range(tw):
        d[:, c] = xp[:, r0:r0 + ar, c * 4:c * 4 + ac]
    V = f32(np.einsum("ik,ucklx,jl->ijucx", btr, d, BT4, optimize=True)).reshape(ar * ac, units * tw, C)
    M = np.matmul(V, U).reshape(ar, ac, units, tw, O).astype(np.float64)
    Y = f32(np.einsum("ik,klnco,jl->ncijo", atr, M, AT4, optimize=True))       # [units][tw][mr][4][O]
    return Y.transpose(0, 2, 1, 3, 4).reshape(units, mr, W, O)


def main():
    import bufferx_amd as bx
    from oracle import oracle as O
    O.lib()
    packed = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    rng = np.random.default_rng(8)
    units = 48
    x = np.abs(rng.standard_normal((units, 3, 140, 16))).astype(np.float32)
    tap = bx.weights.cyl_tap_table()
    for l in range(3):
        L = packed["desc"][l]
        x = O.conv(x, tap, L["W"], L["b"], L["relu"])
    L = packed["desc"][3]
    Wl = np.asarray(L["W"], np.float64)
    w = Wl.reshape(8, 3, 3, 16, 128).transpose(1, 2, 0, 3, 4).reshape(3, 3, 128, 128)
    xm = x.transpose(0, 2, 1, 3).reshape(units, 7, 20, 128)
    xp = np.zeros((units, 9, 22, 128))
    xp[:, 1:8, 1:21] = xm
    xp[:, 1:8, 0] = xm[:, :, 19]
    xp[:, 1:8, 21] = xm[:, :, 0]
    ref = np.zeros((units, 7, 20, 128))
    for kh in range(3):
        for kw in range(3):
            ref += np.einsum("uhwc,co->uhwo", xp[:, kh:kh + 7, kw:kw + 20], w[kh, kw])
    y4 = wino(xm, w, BT4, G4, AT4, 4)
    ym = np.concatenate([wino_rows(xm, w, 0, 4, BT4, G4, AT4), wino_rows(xm, w, 4, 3, BT3, G3, AT3)], 1)
    assert np.array_equal(ym[:, :4], y4[:, :4])          # rows 0..3: the same tiles, the same arithmetic
    rep = {"layer": "Cylindrical_Net layer 3 (128 -> 128), pre-bias / pre-ReLU outputs", "units": units,
           "rms_of_output": float(np.sqrt(np.mean(ref ** 2)))}
    for name, y, rows in (("winograd_F4x4_all_rows", y4, slice(0, 7)), ("winograd_F4x4_rows_4_6", y4, slice(4, 7)),
                          ("mixed_F3x4_rows_4_6", ym, slice(4, 7)), ("mixed_all_rows", ym, slice(0, 7))):
        e = np.abs(y[:, rows].astype(np.float64) - ref[:, rows])
        rep[name] = {"max_abs": float(e.max()), "rms_abs": float(np.sqrt(np.mean(e ** 2)))}
    rep["mixed_over_F4x4_rms_rows_4_6"] = rep["mixed_F3x4_rows_4_6"]["rms_abs"] / rep["winograd_F4x4_rows_4_6"]["rms_abs"]
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
