#!/usr/bin/env python
"""tools/remint_distance_form.py CASE... -- run the REFERENCE's own BufferX.forward (tests/golden/make_golden.py::run_reference) with the
numpy stand-ins of the un-vendored CUDA ops switched to nvcc's contracted distance (BX_REF_DIST_FORM=nvcc_fma, tests/golden/ref_harness.py)
and compare with the committed fixture of the same case (minted with the un-fused form): counts, per-scale mutual sets, consensus set,
RANSAC log, pose.  Needs /root/reference (build container only); ~7 CPU-minutes per real-size case.  One JSON row per case."""
import json
import os
import sys

os.environ["BX_REF_DIST_FORM"] = "nvcc_fma"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np


def main():
    import make_golden as MG
    import bufferx_amd as bx
    assert MG.rh.DIST_FORM == "nvcc_fma"
    for name in sys.argv[1:]:
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        cap = MG.run_reference(name)
        S = int(cap["scales_used"])
        row = {"case": name, "form": "nvcc_fma (reference forward re-run) vs the committed fixture (un-fused)",
               "counts_fma": [int(cap["num_inliers"]), int(cap["num_mutual"]), int(cap["num_inlier_ind"]), S],
               "counts_fixture": [int(g["num_inliers"]), int(g["num_mutual"]), int(g["num_inlier_ind"]), int(g["scales_used"])]}
        md = []
        for i in range(S):
            a = set(zip(cap[f"s{i}_s_mids"].tolist(), cap[f"s{i}_t_mids"].tolist()))
            b = set(zip(g[f"s{i}_s_mids"].tolist(), g[f"s{i}_t_mids"].tolist()))
            md.append(len(a ^ b))
        row["mutual_matches_that_differ_per_scale"] = md
        k = 0
        while f"est{k}_T" in g:
            k += 1
        row["consensus_identical"] = bool(np.array_equal(cap[f"est{k - 1}_inlier_ind"], g[f"est{k - 1}_inlier_ind"]))
        row["ransac_pose_max_abs_diff"] = float(np.abs(cap[f"est{k - 1}_T"] - g[f"est{k - 1}_T"]).max())
        row["pose_diff_deg_m"] = [float(x) for x in bx.synth.pose_difference(np.asarray(cap["pose"], np.float64), np.asarray(g["pose"], np.float64))]
        row["desc_rows_not_bit_equal"] = {f"s{i}_{c}": int((np.abs(cap[f"s{i}_{c}_desc"] - g[f"s{i}_{c}_desc"]).max(1) > 0).sum()) for i in range(S) for c in ("src", "tgt")}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
