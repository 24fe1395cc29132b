#!/usr/bin/env python
"""tools/distance_form_study.py -- what moves if the squared distances of FPS / ball_query / SPT are evaluated the way nvcc builds the
reference's un-vendored CUDA ops (default -fmad=true: fmaf(dz, dz, fmaf(dy, dy, dx * dx))) instead of the contract's un-fused
((dx*dx + dy*dy) + dz*dz)?  (review, round 4, item 5.)  CPU only: the oracle pipeline at the REAL sizes (K = 5000 / P = 1024 / S = 3) on
the inputs of reference-minted fixtures, once per form; one JSON row per case:
  fps_indices_that_differ (of 2 x 5000, and the first position), keypoint SETS that differ, neighbour lists that differ per (scale, cloud)
  (of 5000 keypoints), descriptor rows that differ, mutual matches / consensus members that differ (as correspondences by keypoint
  COORDINATES, since indices shift when FPS differs), RANSAC inliers, pose difference between the forms and of each form to the ground truth.
  python tools/distance_form_study.py [case ...] >> profiles/r05_distance_form.jsonl"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(bx, PL, O, packed, name, form):
    from test_gpu_headline import big_case
    cfg, pair, seed = big_case(bx, name)
    cap = {}
    with O.distance_form(form):
        pose, n_inl, n_mut, n_ind, scales = PL.register_pair(pair["src"], pair["tgt"], packed, cfg, pair["aligned_z"], seed, cap)
    return cfg, pair, cap, np.asarray(pose, np.float64), (n_inl, n_mut, n_ind, scales)


def corr_set(pair, cap, scales):
    """accumulated correspondences as (scale, src xyz bytes, tgt xyz bytes): comparable across runs whose FPS orders differ"""
    K = len(cap["s_fps"])
    acc = []
    for i in range(scales):
        sk, tk = pair["src"][cap["s_fps"]], pair["tgt"][cap["t_fps"]]
        for a, b in zip(cap[f"s{i}_s_mids"], cap[f"s{i}_t_mids"]):
            acc.append((i, sk[a].tobytes(), tk[b].tobytes()))
    return acc


def main():
    import bufferx_amd as bx
    from oracle import pipeline as PL, oracle as O
    packed = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    for name in sys.argv[1:] or ["headline_cfg1", "kitti_cfg2"]:
        cfg, pair, c0, p0, t0 = run(bx, PL, O, packed, name, "unfused")
        _, _, c1, p1, t1 = run(bx, PL, O, packed, name, "nvcc_fma")
        row = {"case": name, "counts_unfused": list(map(int, t0)), "counts_nvcc_fma": list(map(int, t1))}
        fd = {}
        for k in ("s_fps", "t_fps"):
            d = c0[k] != c1[k]
            fd[k] = {"indices_that_differ": int(d.sum()), "of": int(len(d)), "first_position": int(np.argmax(d)) if d.any() else None,
                     "keypoint_sets_differ_by": int(len(set(c0[k].tolist()) ^ set(c1[k].tolist())))}
        row["fps"] = fd
        nb = {}
        for key in sorted(k for k in c0 if k.endswith("idx")):
            if c0[key].shape == c1[key].shape:
                nb[key] = {"lists_that_differ": int((c0[key] != c1[key]).any(1).sum()), "of": int(len(c0[key]))}
        row["neighbour_lists"] = nb
        dr = {}
        for key in sorted(k for k in c0 if k.endswith("desc")):
            if c0[key].shape == c1[key].shape:
                dd = np.abs(c0[key].astype(np.float64) - c1[key]).max(1)
                dr[key] = {"rows_not_bit_equal": int((dd > 0).sum()), "rows_beyond_2e-5": int((dd > 2e-5).sum()), "of": int(len(dd))}
        row["descriptor_rows"] = dr
        a0, a1 = corr_set(pair, c0, t0[3]), corr_set(pair, c1, t1[3])
        row["mutual_matches"] = {"unfused": len(a0), "nvcc_fma": len(a1), "differ": len(set(a0) ^ set(a1))}
        s0 = {a0[j] for j in c0[f"s{t0[3] - 1}_inlier_ind"]}
        s1 = {a1[j] for j in c1[f"s{t1[3] - 1}_inlier_ind"]}
        row["consensus"] = {"unfused": len(s0), "nvcc_fma": len(s1), "differ": len(s0 ^ s1)}
        row["pose_diff_between_forms_deg_m"] = [float(x) for x in bx.synth.pose_difference(p0, p1)]
        row["pose_error_vs_ground_truth_deg_m"] = {"unfused": [float(x) for x in bx.synth.pose_error(p0, pair["T_gt"])],
                                                   "nvcc_fma": [float(x) for x in bx.synth.pose_error(p1, pair["T_gt"])]}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
