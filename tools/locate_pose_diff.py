#!/usr/bin/env python
"""tools/locate_pose_diff.py -- WHERE do the 1e-5-degree pose differences between the oracle and the reference-minted real-size indoor
fixtures come from (review, round 4: "3.0e-5 / 4.2e-5 deg, identical in all conv forms, undocumented where exactly")?  CPU only.
For each indoor fixture: the binary64 RANSAC pose (before refinement) and the float32 refined pose of the oracle pipeline against the
fixture's est<k>_T / pose, then the oracle's refinement started from the FIXTURE's RANSAC pose on the fixture's correspondences."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bufferx_amd as bx
    from oracle import pipeline as PL
    from test_gpu_headline import big_case, golden_path
    packed = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    out = []
    for name in sys.argv[1:] or ["headline_cfg1", "headline_cfg1_c", "headline_cfg1_b"]:
        g = np.load(golden_path(name))
        cfg, pair, seed = big_case(bx, name)
        cap = {}
        pose, n_inl, n_mut, n_ind, scales = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed, cap)
        k = 0
        while f"est{k}_T" in g:
            k += 1
        d_init = bx.synth.pose_difference(np.asarray(cap["init_pose"], np.float64), np.asarray(g[f"est{k - 1}_T"], np.float64))
        d_final = bx.synth.pose_difference(np.asarray(pose, np.float64), np.asarray(g["pose"], np.float64))
        row = dict(case=name, ransac_pose_diff_deg_m=[float(d_init[0]), float(d_init[1])], refined_pose_diff_deg_m=[float(d_final[0]), float(d_final[1])],
                   ransac_max_abs_entry_diff=float(np.abs(np.asarray(cap["init_pose"], np.float64) - g[f"est{k - 1}_T"]).max()),
                   refined_max_abs_entry_diff=float(np.abs(np.asarray(pose, np.float64) - g["pose"]).max()))
        print(json.dumps(row))
        out.append(row)
    return out


if __name__ == "__main__":
    main()
