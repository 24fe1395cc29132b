"""oracle/pipeline.py -- TEST INFRASTRUCTURE: the whole inference hot path chained on the CPU oracle.

Mirrors BufferX.forward's inference branch (reference models/BUFFERX.py:257-467), MiniSpinNet.forward
(models/patch_embedder.py:44-90) and PoseEstimator.estimate_pose (models/pose_estimator.py:22-117),
stage by stage, on the C oracle.  Randomness is explicit (seed): permutation stream = 2*scale + cloud,
RANSAC call c uses mix64(seed, 0x5AC0000 + c), the >200k-point subsample of the radius estimation
uses stream 0x200 (the reference uses unseeded torch.randint, models/BUFFERX.py:664-665).
"""
import numpy as np
from . import oracle as O

def _arith(cfg):
    import bufferx_amd.config as C        # pure-python knob table (no GPU code)
    return C.arith_of(cfg)


def _w():
    import bufferx_amd.weights as W  # pure-python weight packing / tap tables (no GPU code)
    return W


def desc_forward(cloud, kpts, des_r, aligned, perm, pw, cfg, cap=None, tag=""):
    W = _w()
    P = cfg.patch.num_points_per_patch
    pts_perm = cloud[perm]
    idx, patches = O.ball_group(pts_perm, kpts, np.float32(des_r), P)
    R, feat = O.patch_features(patches, des_r, aligned, pw["pnt_w"], pw["pnt_b"], cfg.patch.rad_n, cfg.patch.ele_n,
                               cfg.patch.azi_n, cfg.patch.voxel_sample, cfg.patch.delta)
    tap = W.cyl_tap_table(cfg.patch.ele_n, cfg.patch.azi_n)
    x = feat
    for L in pw["desc"]:
        x = O.desc_conv(x, tap, L["W"], L["b"], L["relu"], form=_arith(cfg)["desc_conv"])     # cfg.arith: the form the product is configured for
    desc, equi = O.desc_head(x, pw["pool_w1"], pw["pool_b1"], pw["pool_w2"], pw["pool_b2"])
    if cap is not None:
        cap[tag + "idx"] = idx
        cap[tag + "patches"] = patches
        cap[tag + "R"] = R
        cap[tag + "feat"] = feat
        cap[tag + "x"] = x
        cap[tag + "desc"] = desc
        cap[tag + "equi"] = equi
    return desc, equi, R.reshape(-1, 9)


def pose_forward(s_equi, t_equi, s_mids, t_mids, pw, cfg, cap=None, tag=""):
    W = _w()
    layers = list(zip(pw["pose"], W.pose_geometry(cfg.patch.ele_n, cfg.patch.azi_n)))
    ar = _arith(cfg)
    if ar["cost_l0"] == "collapsed":
        # layer 0 on the implicit cost volume: the collapsed binary64 form (bxo_cost_l0); layers 1..9 are explicit convolutions
        x = O.cost_l0(s_equi, t_equi, s_mids, t_mids, pw["pose"][0]["W"], pw["pose"][0]["b"], cfg.patch.ele_n, cfg.patch.azi_n)
        layers = layers[1:]
    else:
        x = O.cost_volume(s_equi, t_equi, s_mids, t_mids, cfg.patch.ele_n, cfg.patch.azi_n)
    first = len(pw["pose"]) - len(layers)
    for li, (L, (dims, k, out)) in enumerate(layers):
        tap, _ = W.valid_tap_table(dims, k)
        x = O.pose_conv(first + li, x, tap, dims, L["W"], L["b"], L["relu"], form=ar["pose_conv"])
    ind = O.soft_argmax(x, cfg.patch.azi_n)
    if cap is not None:
        cap[tag + "logits"] = x
        cap[tag + "ind"] = ind
    return ind


def estimate_pose(ss, tt, inlier_ind, cfg, seed, call):
    if cfg.match.get("pose_estimator", "ransac") == "kiss_matcher":      # models/pose_estimator.py:43-44
        T, info = O.kiss_solve(ss, tt, inlier_ind, cfg.match.get("kiss_resolution", 0.3))
        return T, int(info[0]), int(info[3])
    T, n, it = O.ransac(ss, tt, inlier_ind, cfg.match.dist_th, cfg.match.similar_th, cfg.match.confidence,
                        cfg.match.iter_n, O.mix64(seed, 0x5AC0000 + call))
    return T, n, it


def register_pair(src, tgt, pw, cfg, aligned, seed, cap=None, perms=None):
    """Returns (pose 4x4, num_inliers, num_mutual, num_inlier_ind, scales_used). pw = weights.fold_and_pack(sd).
    perms = (perm_src [S][n_src], perm_tgt [S][n_tgt]) replaces the seed-derived permutations (a caller that drew them itself)."""
    src = np.ascontiguousarray(src, np.float32)
    tgt = np.ascontiguousarray(tgt, np.float32)
    K = cfg.patch.num_fps
    nk = cfg.patch.num_points_radius_estimate
    S = cfg.patch.num_scales
    s_idx = O.fps(src, max(K, nk))
    t_idx = O.fps(tgt, max(K, nk))
    kpts1, kpts2 = src[s_idx[:nk]], tgt[t_idx[:nk]]
    src_kpts, tgt_kpts = src[s_idx[:K]], tgt[t_idx[:K]]
    if cap is not None:
        cap["s_fps"], cap["t_fps"] = s_idx, t_idx
    R_acc, t_acc, ss_acc, tt_acc = [], [], [], []
    early = bool(cfg.match.get("enable_early_exit", True))
    pose, num_inliers, calls, should_exit, scales_used = None, 0, 0, False, 0
    inlier_ind = np.zeros(0, np.int32)
    for i in range(S):
        big, bk = (src, kpts1) if len(src) > len(tgt) else (tgt, kpts2)
        n_orig = len(big)
        pts = big
        if n_orig > 200000:
            sel = np.array([O.mix64(seed, (0x200 << 32) + j) % n_orig for j in range(200000)], np.int64)
            pts = big[sel]
        des_r = O.radius(pts, n_orig, bk, cfg.patch.search_radius_thresholds[i])
        if cap is not None:
            cap[f"s{i}_des_r"] = des_r
        perm_s = O.make_perm(len(src), seed, 2 * i) if perms is None else np.asarray(perms[0][i])
        perm_t = O.make_perm(len(tgt), seed, 2 * i + 1) if perms is None else np.asarray(perms[1][i])
        s_desc, s_equi, s_R = desc_forward(src, src_kpts, des_r, aligned, perm_s, pw, cfg, cap, f"s{i}_src_")
        t_desc, t_equi, t_R = desc_forward(tgt, tgt_kpts, des_r, aligned, perm_t, pw, cfg, cap, f"s{i}_tgt_")
        s_mids, t_mids, _, _ = O.mutual(s_desc, t_desc)
        ind = pose_forward(s_equi, t_equi, s_mids, t_mids, pw, cfg, cap, f"s{i}_")
        R, t = O.hypotheses(ind, s_R[s_mids], t_R[t_mids], src_kpts[s_mids], tgt_kpts[t_mids], cfg.patch.azi_n)
        R_acc.append(R); t_acc.append(t); ss_acc.append(src_kpts[s_mids]); tt_acc.append(tgt_kpts[t_mids])
        scales_used = i + 1
        Rc, tc = np.concatenate(R_acc), np.concatenate(t_acc)
        ss, tt = np.concatenate(ss_acc), np.concatenate(tt_acc)
        inlier_ind, best, counts = O.consensus(Rc, tc, ss, tt, cfg.match.inlier_th, cfg.patch.azi_n)
        if cap is not None:
            cap[f"s{i}_s_mids"], cap[f"s{i}_t_mids"] = s_mids, t_mids
            cap[f"s{i}_R"], cap[f"s{i}_t"] = R, t
            cap[f"s{i}_inlier_ind"], cap[f"s{i}_best"] = inlier_ind, best
        if early and i == 0:
            pose, num_inliers, _ = estimate_pose(ss, tt, inlier_ind, cfg, seed, calls)
            calls += 1
            should_exit = num_inliers >= cfg.match.get("early_exit_min_inliers", 15)
            if should_exit:
                break
    if (not early) or (early and not should_exit):
        pose, num_inliers, _ = estimate_pose(ss, tt, inlier_ind, cfg, seed, calls)
        calls += 1
    if cap is not None:
        cap["init_pose"] = pose.copy()
    if cfg.test.pose_refine is True:
        T, _ = O.refine(ss, tt, cfg.match.dist_th, pose.astype(np.float32))
        pose = T
    return pose, int(num_inliers), int(len(ss)), int(len(inlier_ind)), scales_used
