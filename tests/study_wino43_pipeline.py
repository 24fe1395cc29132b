"""Numerical study (CPU, not a test): the reference-minted small fixtures through the ORACLE pipeline with the six >= 64-channel
Cylindrical_Net layers replaced by an fp32 emulation of Winograd F(4x4, 3x3) (tests/study_wino43_error.py::wino).  Question: would a
kernel with 1.8x fewer multiplications than the shipped F(2x2, 3x3) form keep counts / poses within the reference tolerance?
Run:  python tests/study_wino43_pipeline.py [case ...]   (JSON lines)."""
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from study_wino43_error import wino, BT4, G4, AT4      # noqa: E402
from test_gpu_pipeline import CASES, make_case           # noqa: E402


def main():
    import bufferx_amd as bx
    from oracle import oracle as O
    from oracle import pipeline as PL
    O.lib()
    packed = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    exact = O.desc_conv
    state = {"on": False, "maxd": 0.0}

    def desc_conv43(x, tap, W, bias, relu):
        W = np.asarray(W)
        if not state["on"] or W.shape[-1] < 64:
            return exact(x, tap, W, bias, relu)
        units, nch = x.shape[0], x.shape[1]
        cin, cout = nch * 16, W.shape[-1]
        w = np.asarray(W, np.float64).reshape(nch, 3, 3, 16, cout).transpose(1, 2, 0, 3, 4).reshape(3, 3, cin, cout)
        out = np.zeros((units, cout // 16, 140, 16), np.float32)
        for u0 in range(0, units, 512):
            xm = np.asarray(x[u0:u0 + 512], np.float32).transpose(0, 2, 1, 3).reshape(-1, 7, 20, cin)
            y = wino(xm, w, BT4, G4, AT4, 4) + np.asarray(bias, np.float32)[None, None, None, :]
            if relu:
                y = np.maximum(y, 0.0)
            out[u0:u0 + 512] = y.astype(np.float32).reshape(-1, 140, cout // 16, 16).transpose(0, 2, 1, 3)
        return out

    O.desc_conv = desc_conv43
    names = sys.argv[1:] or ["indoor_success", "indoor_3scale", "baseline_cfg0", "outdoor_3scale"]
    for name in names:
        big = name in ("headline_cfg1", "kitti_cfg2", "tiers_early")
        if big:
            from test_gpu_headline import big_case
            cfg, pair, seed = big_case(bx, name)
        else:
            cfg, pair, seed = make_case(bx, name)
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        rep = {"case": name}
        if not big:
            state["on"] = False
            r0 = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed)
            rep["counts_shipped_form"] = [int(v) for v in r0[1:]]
        state["on"] = True
        cap = {}
        r1 = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed, cap)
        rre_g, rte_g = bx.synth.pose_difference(np.asarray(r1[0], np.float64), g["pose"])
        rep["counts_F4x4"] = [int(v) for v in r1[1:]]
        rep["counts_reference"] = [int(g["num_inliers"]), int(g["num_mutual"]), int(g["num_inlier_ind"]), int(g["scales_used"])]
        rep["pose_vs_reference_deg_m"] = [float(rre_g), float(rte_g)]
        if not big:
            rre_e, rte_e = bx.synth.pose_difference(np.asarray(r1[0], np.float64), np.asarray(r0[0], np.float64))
            rep["pose_vs_shipped_form_deg_m"] = [float(rre_e), float(rte_e)]
        else:                                         # the reference's own mutual / consensus sets and sampled descriptor rows
            rs = int(g["row_stride"])
            for i in range(int(r1[4])):
                same = bool(np.array_equal(cap[f"s{i}_s_mids"], g[f"s{i}_s_mids"]) and np.array_equal(cap[f"s{i}_t_mids"], g[f"s{i}_t_mids"]))
                dmax = max(float(np.abs(cap[f"s{i}_{c}_desc"][::rs].astype(np.float64) - g[f"s{i}_{c}_desc"]).max()) for c in ("src", "tgt"))
                rep[f"scale{i}"] = {"mutual_F4x4": int(len(cap[f"s{i}_s_mids"])), "mutual_ref": int(len(g[f"s{i}_s_mids"])), "mutual_sets_identical": same,
                                    "desc_max_abs_diff_vs_reference_rows": dmax}
            k = 0
            while f"est{k}_T" in g:
                k += 1
            rep["consensus_set_identical"] = bool(np.array_equal(cap[f"s{int(r1[4]) - 1}_inlier_ind"], g[f"est{k - 1}_inlier_ind"]))
        print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
