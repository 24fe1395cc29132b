"""MEASUREMENT of a split-precision convolution (round-2 review, item 8: "measure -- do not ship"): the 128 -> 128 Cylindrical_Net
layer on the bf16 matrix cores with every fp32 operand cut into three bf16 pieces and the six leading partial products accumulated
in fp32 (buffer-x_amd/csrc/k_split.hip, BX_EXP_SPLIT_CONV=1).  The shipped path stays exact f32; these tests put numbers on what the
split form would do: per-element error against a binary64 convolution next to the two exact-f32 forms (direct, Winograd),
descriptor deltas, and count / des_r / pose deltas on the reference-minted fixtures.  The report goes to $BX_SPLIT_REPORT (JSON
lines) when set; tools/gpu_r3m.sh copies it to profiles/.  The real-size fixtures are covered by running
tests/test_gpu_headline.py::test_headline_vs_reference under BX_EXP_SPLIT_CONV=1 (same script)."""
import json
import os
import numpy as np
import pytest

from test_gpu_pipeline import CASES, make_case, run_gpu

pytestmark = pytest.mark.gpu


def _report(obj):
    path = os.environ.get("BX_SPLIT_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(obj) + "\n")


def _np(t):
    return t.detach().cpu().numpy()


def _ctx(bx, packed, monkeypatch, desc_conv, split):
    from bufferx_amd import lib
    monkeypatch.setenv("BX_DESC_CONV", desc_conv)
    monkeypatch.setenv("BX_EXP_SPLIT_CONV", "1" if split else "0")
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 256, 128, 1
    cfg.patch.search_radius_thresholds = [5]
    cfg.patch.num_points_radius_estimate = 256
    return lib.Context(cfg, max_points=20000, device=0, packed_weights=packed)


def test_split_layer_error_against_binary64(bx, packed, monkeypatch):
    """Layer 3 (128 -> 128) on realistic activations (the output of the exact layers 0..2 on |N(0,1)| features): error of the direct
    f32, the Winograd f32 and the split-bf16 forms against the same convolution in binary64."""
    from bufferx_amd import lib
    L = packed["desc"][3]
    rng = np.random.default_rng(8)
    units = 96
    feat = np.abs(rng.standard_normal((units, 3, 140, 16))).astype(np.float32)
    outs = {}
    x3 = None
    for name, (dc, sp) in {"direct_f32": ("direct", False), "winograd_f32": ("winograd", False), "split_bf16x3": ("direct", True)}.items():
        c = _ctx(bx, packed, monkeypatch, dc, sp)
        if x3 is None:                                           # the layer's input: exact direct layers 0..2
            x = lib.logical_to_chunked(feat)
            for l in range(3):
                Ll = packed["desc"][l]
                x = c.conv_layer(0, l, x, (units, Ll["W"].shape[-1] // 16, 140, 16))
            x3 = x
        outs[name] = lib.chunked_to_logical(_np(c.conv_layer(0, 3, x3, (units, 8, 140, 16)))).astype(np.float64)
        c.close()
    xin = lib.chunked_to_logical(_np(x3)).astype(np.float64)      # [units][8][140][16]
    tap = bx.weights.cyl_tap_table()
    W = np.asarray(L["W"], np.float64)                            # [8][9][16][128]
    ref = np.zeros((units, 140, 128))
    for t in range(9):
        src = tap[t]
        ok = src >= 0
        g = np.zeros((units, 8, 140, 16))
        g[:, :, ok, :] = xin[:, :, src[ok], :]
        ref += np.einsum("ucpk,cko->upo", g, W[:, t])
    ref += np.asarray(L["b"], np.float64)[None, None, :]
    ref = np.maximum(ref, 0.0)
    ref_l = ref.reshape(units, 140, 8, 16).transpose(0, 2, 1, 3)  # logical [units][8][140][16]
    scale = float(np.sqrt(np.mean(ref_l ** 2)))
    rep = {"test": "layer3_error_vs_binary64", "units": units, "rms_of_output": scale, "max_of_output": float(ref_l.max())}
    for name, y in outs.items():
        e = np.abs(y - ref_l)
        rep[name] = {"max_abs": float(e.max()), "rms_abs": float(np.sqrt(np.mean(e ** 2))), "max_abs_over_rms_output": float(e.max() / scale)}
    d = np.abs(outs["split_bf16x3"] - outs["direct_f32"])
    rep["split_vs_direct"] = {"max_abs": float(d.max()), "bit_identical_share": float(np.mean(d == 0))}
    _report(rep)
    # sanity: all three are fp32-grade; the split form is no worse than 4x the direct form's own rounding error
    assert rep["direct_f32"]["max_abs_over_rms_output"] < 1e-5
    assert rep["split_bf16x3"]["max_abs"] <= 4 * max(rep["direct_f32"]["max_abs"], rep["winograd_f32"]["max_abs"])


def test_split_descriptor_delta(bx, packed, monkeypatch):
    """The whole descriptor network (8 layers + head) with layer 3 in the split form against the shipped form."""
    from bufferx_amd import lib
    rng = np.random.default_rng(5)
    K = 256                                                      # = num_fps of the test context
    feat = lib.logical_to_chunked(np.abs(rng.standard_normal((K, 3, 140, 16))).astype(np.float32))
    res = {}
    for name, sp in {"shipped": False, "split": True}.items():
        c = _ctx(bx, packed, monkeypatch, "winograd", sp)
        desc, equi, _ = c.desc_net(feat)
        res[name] = (_np(desc).astype(np.float64), _np(equi).astype(np.float64))
        c.close()
    dd = np.abs(res["split"][0] - res["shipped"][0])
    de = np.abs(res["split"][1] - res["shipped"][1])
    rep = {"test": "descriptor_delta", "patches": K, "desc_max_abs_delta": float(dd.max()), "desc_rms_delta": float(np.sqrt(np.mean(dd ** 2))),
           "desc_rms": float(np.sqrt(np.mean(res["shipped"][0] ** 2))), "equi_max_abs_delta": float(de.max())}
    _report(rep)
    assert rep["desc_max_abs_delta"] < 1e-4


@pytest.mark.parametrize("name", list(CASES))
def test_split_on_reference_fixtures(bx, packed, oracle, golden_dir, monkeypatch, name):
    """Every reference-minted small fixture with layer 3 in the split form: counts, des_r and pose against the fixture and against the
    shipped (exact f32) run."""
    cfg, pair, seed = make_case(bx, name)
    monkeypatch.setenv("BX_EXP_SPLIT_CONV", "0")
    pose0, inl0, mut0, ind0, sc0, desr0 = run_gpu(bx, packed, oracle, cfg, pair, seed)
    monkeypatch.setenv("BX_EXP_SPLIT_CONV", "1")
    pose1, inl1, mut1, ind1, sc1, desr1 = run_gpu(bx, packed, oracle, cfg, pair, seed)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    rre_g, rte_g = bx.synth.pose_difference(pose1, g["pose"])
    rre_e, rte_e = bx.synth.pose_difference(pose1, pose0)
    rep = {"test": "fixture", "name": name,
           "counts_split": [int(inl1), int(mut1), int(ind1), int(sc1)], "counts_exact": [int(inl0), int(mut0), int(ind0), int(sc0)],
           "counts_reference": [int(g["num_inliers"]), int(g["num_mutual"]), int(g["num_inlier_ind"]), int(g["scales_used"])],
           "des_r_max_delta_vs_exact": float(np.max(np.abs(np.asarray(desr1[:sc1]) - np.asarray(desr0[:sc1])))),
           "pose_vs_reference_deg_m": [float(rre_g), float(rte_g)], "pose_vs_exact_deg_m": [float(rre_e), float(rte_e)]}
    _report(rep)
    assert rre_g < 1e-4 and rte_g < 1e-4                          # the north_star tolerance, against the reference's own output
