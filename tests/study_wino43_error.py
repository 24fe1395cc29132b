"""Numerical study (CPU, not a test): how much accuracy would a Winograd F(4x4, 3x3) form of the 128 -> 128 Cylindrical_Net layer
cost?  F(4x4, 3x3) needs 36 multiplications per 16 outputs (F(2x2, 3x3): 16 per 4; direct: 9 per 1) but its transform matrices hold
non-dyadic constants up to 8 / down to 1/24, so intermediate values are amplified and cancel in the output transform.  The script
emulates the fp32 pipeline (input transform, per-plane channel contraction, output transform all rounded to fp32; filter transform in
binary64 rounded once, like the shipped F(2x2) form) on realistic activations (the exact layers 0..2 of the descriptor stack on
|N(0,1)| features, through the oracle) and reports the error against a binary64 convolution next to the F(2x2, 3x3) form and the direct
fp32 form.  Run:  python tests/study_wino43_error.py   (prints one JSON object; LABBOOK.md section 2 quotes it)."""
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
              np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def f32(x):
    return np.asarray(x, np.float32)


def wino(x, w, bt, g, at, m):
    """x [units][H][W][C] fp32 (cylindrical: circular in W, zero in H), w [3][3][C][O] binary64.  fp32 emulation: every stage rounded
    to fp32 (matrix products of the transforms are evaluated in binary64 and rounded once per stage, which is slightly optimistic for the
    transforms; the channel contraction accumulates in fp32 through numpy's float32 matmul)."""
    units, H, W, C = x.shape
    O = w.shape[-1]
    a = m + 2
    th, tw = -(-H // m), W // m
    assert W % m == 0
    xp = np.zeros((units, th * m + 2, W + 2, C), np.float32)
    xp[:, 1:H + 1, 1:W + 1] = x
    xp[:, 1:H + 1, 0] = x[:, :, W - 1]
    xp[:, 1:H + 1, W + 1] = x[:, :, 0]
    U = f32(np.einsum("ik,klco,jl->ijco", g, w, g)).reshape(a * a, C, O)     # binary64 -> fp32 once
    out = np.zeros((units, th * m, W, O), np.float32)
    for u0 in range(0, units, 128):                                           # all tiles of 128 units at once
        xb = xp[u0:u0 + 128]
        nb = xb.shape[0]
        d = np.empty((nb, th, tw, a, a, C), np.float64)
        for r in range(th):
            for c in range(tw):
                d[:, r, c] = xb[:, r * m:r * m + a, c * m:c * m + a]
        V = f32(np.einsum("ik,urcklx,jl->ijurcx", bt, d, bt, optimize=True)).reshape(a * a, nb * th * tw, C)    # fp32 V
        M = np.matmul(V, U)                                                   # fp32 (BLAS) accumulation, one GEMM per plane
        M = M.reshape(a, a, nb, th, tw, O).astype(np.float64)
        Y = f32(np.einsum("ik,klurco,jl->urcijo", at, M, at, optimize=True))  # fp32 Y [nb][th][tw][m][m][O]
        out[u0:u0 + nb] = Y.transpose(0, 1, 3, 2, 4, 5).reshape(nb, th * m, W, O)
    return out[:, :H]


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
        x = O.conv(x, tap, L["W"], L["b"], L["relu"])                                 # direct fp32 form of layers 0..2
    L = packed["desc"][3]
    Wl = np.asarray(L["W"], np.float64)                                               # [8][9][16][128]
    w = Wl.reshape(8, 3, 3, 16, 128).transpose(1, 2, 0, 3, 4).reshape(3, 3, 128, 128)
    xm = x.transpose(0, 2, 1, 3).reshape(units, 7, 20, 128)                           # [units][H][W][C]
    # binary64 reference
    xp = np.zeros((units, 9, 22, 128))
    xp[:, 1:8, 1:21] = xm
    xp[:, 1:8, 0] = xm[:, :, 19]
    xp[:, 1:8, 21] = xm[:, :, 0]
    ref = np.zeros((units, 7, 20, 128))
    for kh in range(3):
        for kw in range(3):
            ref += np.einsum("uhwc,co->uhwo", xp[:, kh:kh + 7, kw:kw + 20], w[kh, kw])
    direct = O.conv(x, tap, L["W"], np.zeros_like(L["b"]), False).transpose(0, 2, 1, 3).reshape(units, 7, 20, 128)
    y2 = wino(xm, w, BT2, G2, AT2, 2)
    y4 = wino(xm, w, BT4, G4, AT4, 4)
    rms = float(np.sqrt(np.mean(ref ** 2)))
    rep = {"layer": "Cylindrical_Net layer 3 (128 -> 128), pre-bias / pre-ReLU outputs", "units": units, "rms_of_output": rms,
           "max_of_output": float(np.abs(ref).max())}
    for name, y in (("direct_f32", direct), ("winograd_F2x2_f32", y2), ("winograd_F4x4_f32", y4)):
        e = np.abs(y.astype(np.float64) - ref)
        rep[name] = {"max_abs": float(e.max()), "rms_abs": float(np.sqrt(np.mean(e ** 2)))}
    rep["F4x4_over_F2x2_rms"] = rep["winograd_F4x4_f32"]["rms_abs"] / rep["winograd_F2x2_f32"]["rms_abs"]
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
