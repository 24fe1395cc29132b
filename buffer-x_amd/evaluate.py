"""The step behind the hot path: per-pair metrics, the 3DMatch .log writer and the RMSE-recall evaluator,
batched for pairs that were sharded over ranks.

    compute_rte / compute_rre            <- utils/SE3.py:134-165
    write_3dmatch_logs                   <- test.py:150-165 (one "a+" open per pair there; here ONE write per scene, same bytes)
    loadlog / read_trajectory / read_trajectory_info / computeTransformationErr / evaluate_registration
                                         <- utils/tools.py:49-129 (nibabel.quaternions.mat2quat restated in mat2quat below)
    summarize                            <- test.py:255-270, 327-338 (recall, RTE / RRE mean +- std, inlier statistics, timings with
                                            the first FIRST_A_FEW_FRAMES = 5 pairs excluded)
    save_per_sample_results              <- utils/result_io.py:7-50
    pack_state / gather_states           one float64 row per pair and ONE all-gather (dist.gather_rows) bring every rank's pairs to
                                         rank 0, which then writes and evaluates in one pass -- at >= 200 pairs/s the reference's
                                         per-pair open/append/close is the serial bottleneck.

Same names, argument meaning and return values as the reference functions; evaluate_registration is vectorised over the pairs."""
import csv
import math
import os

import numpy as np

from . import dist as _dist

FIRST_A_FEW_FRAMES = 5          # test.py:24
STATE_W = 32                    # float64 row: id, success, rte, rre, 4 counts, 5 times, pose dtype flag, pose[16], spare


def compute_rte(trans_est, trans_gt):
    """|t_est - t_gt| (utils/SE3.py:134-147)"""
    d = trans_est[:3, 3] - trans_gt[:3, 3]
    return np.linalg.norm(d)


def compute_rre(trans_est, trans_gt):
    """angle of R_est^T R_gt in degrees, cosine clipped to +-(1 - 1e-16) (utils/SE3.py:150-165)"""
    c = (np.trace(trans_est[:3, :3].T @ trans_gt[:3, :3]) - 1) / 2
    lim = 1 - 1e-16
    return np.arccos(np.clip(c, -lim, lim)) * 180 / math.pi


def mat2quat(M):
    """nibabel.quaternions.mat2quat (Bar-Itzhack 2000): eigenvector of the largest eigenvalue of the symmetric 4x4 K built from the
    rotation matrix, returned as (w, x, y, z) with w >= 0."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, np.float64).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = q * -1
    return q


def computeTransformationErr(trans, info):
    t, r = trans[:3, 3], trans[:3, :3]
    er = np.concatenate([t, mat2quat(r)[1:]], axis=0)
    return (er.reshape(1, 6) @ info @ er.reshape(6, 1) / info[0, 0]).item()


def _blocks(filename, rows):
    """A trajectory file is a sequence of blocks: one tab-separated header line followed by `rows` tab-separated matrix rows."""
    with open(filename) as f:
        lines = f.readlines()
    nb = (len(lines) + rows) // (rows + 1)
    return [lines[b * (rows + 1):(b + 1) * (rows + 1)] for b in range(nb)]


def loadlog(gtpath):
    """{"<i>_<j>": float64 [4,4]} from <gtpath>/gt.log (utils/tools.py:49-64; the 3DMatch loader inverts these into relt_pose,
    dataset/threedmatch.py:123)."""
    out = {}
    for blk in _blocks(os.path.join(gtpath, "gt.log"), 4):
        if len(blk) < 5:
            break
        head = blk[0].replace("\n", "").split("\t")[0:3]
        T = np.zeros([4, 4])
        for r in range(4):
            T[r] = [float(x) for x in blk[1 + r].replace("\n", "").split("\t")[0:4]]
        out[f"{int(head[0])}_{int(head[1])}"] = T
    return out


def read_trajectory(filename, dim=4):
    """-> (keys [n,3] array of strings, float32 [n,dim,dim]); same parse as utils/tools.py:67-77 (first `dim` tab fields of a row)."""
    keys, mats = [], []
    for blk in _blocks(filename, dim):
        head = blk[0].split("\t")
        keys.append([head[0].strip(), head[1].strip(), head[2].strip()])
        mats.extend(row.split("\t")[0:dim] for row in blk[1:])
    return np.asarray(keys), np.asarray(mats, dtype=np.float32).reshape(-1, dim, dim)


def read_trajectory_info(filename, dim=6):
    """-> (number of fragments from the first header line, float32 [n,dim,dim] information matrices) (utils/tools.py:80-96)."""
    blocks = [b for b in _blocks(filename, dim) if len(b) == dim + 1]
    infos = np.asarray([[np.array(row.split(), np.float64) for row in blk[1:]] for blk in blocks], dtype=np.float32)
    with open(filename) as f:
        n_frag = int(f.readline().strip().split()[2])
    return n_frag, infos.reshape(-1, dim, dim)


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2=0.2):
    """utils/tools.py:105-129, vectorised: one batched inverse / product and one quaternion per evaluated pair."""
    err2 = err2 ** 2
    gt_mask = np.zeros((num_fragment, num_fragment), dtype=np.int64)
    gi = gt_pairs[:, 0].astype(np.int64) if len(gt_pairs) else np.zeros(0, np.int64)
    gj = gt_pairs[:, 1].astype(np.int64) if len(gt_pairs) else np.zeros(0, np.int64)
    for idx in np.nonzero(gj - gi > 1)[0]:          # later entries overwrite earlier ones, as in the reference loop
        gt_mask[gi[idx], gj[idx]] = idx
    n_gt = np.sum(gt_mask > 0)
    n = result_pairs.shape[0]
    transformation_errors = np.full(n, np.nan)
    ri, rj = result_pairs[:, 0].astype(np.int64), result_pairs[:, 1].astype(np.int64)
    gidx = gt_mask[ri, rj] if n else np.zeros(0, np.int64)
    sel = np.nonzero(gidx > 0)[0]
    flags = np.full(n, 2, np.int64)
    if len(sel):
        rel = np.linalg.inv(gt[gidx[sel]]) @ result[sel]
        for k, idx in enumerate(sel):
            transformation_errors[idx] = computeTransformationErr(rel[k], gt_info[gidx[idx]])
        flags[sel] = np.where(transformation_errors[sel] <= err2, 0, 1)
    good = int(np.sum(flags == 0))
    n_res = len(sel)
    return good / max(n_res, 1e-6), good / n_gt, [int(f) for f in flags], transformation_errors


# ---------------------------------------------------------------------------------------------------- 3DMatch log writer
def _log_block(src_id, tgt_id, trans_est):
    trans = np.linalg.inv(trans_est)                # dtype follows the pose (float64 from RANSAC, float32 after refinement)
    s = f"{src_id}\t {tgt_id}\t  1\n"
    for r in range(4):
        s += f"{trans[r, 0]}\t {trans[r, 1]}\t {trans[r, 2]}\t {trans[r, 3]}\t \n"
    return s


def write_3dmatch_logs(benchmark, timestr, entries, root="."):
    """entries: iterable of (src_id_path, tgt_id_path, trans_est) in dataset order, ids as data_source["src_id"] holds them
    (".../<scene>/cloud_bin_<k>").  Produces logs/log_<benchmark>/<scene>/<timestr>.log with the bytes the reference appends pair
    by pair, in ONE write per scene.  Returns {scene: path}."""
    per_scene = {}
    for src_path, tgt_path, trans_est in entries:
        scene = src_path.split("/")[-2]
        src_id = src_path.split("/")[-1].split("_")[-1]
        tgt_id = tgt_path.split("/")[-1].split("_")[-1]
        trans_est = trans_est if trans_est is not None else np.eye(4)
        per_scene.setdefault(scene, []).append(_log_block(src_id, tgt_id, trans_est))
    out = {}
    for scene, blocks in per_scene.items():
        logpath = os.path.join(root, f"logs/log_{benchmark}/{scene}")
        os.makedirs(logpath, exist_ok=True)
        path = os.path.join(logpath, f"{timestr}.log")
        with open(path, "a+") as f:
            f.write("".join(blocks))
        out[scene] = path
    return out


# ---------------------------------------------------------------------------------------------------- per-pair states
def pack_state(pair_id, trans_est, trans_gt, num_inliers, num_mutual_inliers, num_inlier_ind, scales_used, data_time_s, model_time_s,
               times, rte_thresh, rre_thresh):
    """One float64 row per pair: [id, success, rte, rre, 4 counts, data_t, model_t, desc_t, pose_t, poseest_t, pose dtype (32 / 64),
    pose[16]] -- the `states` row of test.py:175-188 plus what the log writer needs."""
    trans_est = trans_est if trans_est is not None else np.eye(4)
    rte = compute_rte(trans_est, trans_gt)
    rre = compute_rre(trans_est, trans_gt)
    row = np.zeros(STATE_W)
    row[0] = pair_id
    row[1] = float(rte < rte_thresh and rre < rre_thresh)
    row[2], row[3] = rte, rre
    row[4:8] = [num_inliers, num_mutual_inliers, num_inlier_ind, scales_used]
    row[8], row[9] = data_time_s, model_time_s
    row[10:13] = list(times)[:3]
    row[13] = 32 if np.asarray(trans_est).dtype == np.float32 else 64
    row[14:30] = np.asarray(trans_est, np.float64).reshape(-1)
    return row


def state_pose(row):
    return np.asarray(row[14:30]).reshape(4, 4).astype(np.float32 if row[13] == 32 else np.float64)


def gather_states(local_rows, n_pairs, device=None):
    """Every rank's rows -> all rows ordered by pair id (the ONE collective of the evaluation)."""
    return _dist.gather_rows(np.asarray(local_rows, np.float64).reshape(-1, STATE_W), n_pairs, device)


def states_matrix(rows):
    """-> the [n, 12] `states` array of test.py:255 (success, rte, rre, 4 counts, 5 times)."""
    rows = np.asarray(rows)
    return rows[:, 1:13].copy()


def summarize(states):
    states = np.asarray(states)
    ok = states[:, 0] == 1
    out = dict(recall=states[:, 0].sum() / states.shape[0],
               rte_mean=states[ok, 1].mean(), rre_mean=states[ok, 2].mean(), rte_std=states[ok, 1].std(), rre_std=states[ok, 2].std(),
               inliers_mean=states[:, 3].mean(), inliers_std=states[:, 3].std(),
               mutual_inliers_mean=states[:, 4].mean(), mutual_inliers_std=states[:, 4].std(),
               inlier_ind_mean=states[:, 5].mean(), inlier_ind_std=states[:, 5].std(),
               scales_used_mean=states[:, 6].mean(), scales_used_std=states[:, 6].std())
    all_times = states[:, 7:12]
    eff = all_times[FIRST_A_FEW_FRAMES:] if len(all_times) > FIRST_A_FEW_FRAMES else all_times
    out["average_times"] = eff.mean(axis=0)
    out["std_times"] = eff.std(axis=0)
    return out


_CSV_HEAD = ("sample_id success rte_m rre_deg num_inliers num_mutual_inliers num_inlier_ind scales_used data_time_s model_time_s "
             "desc_time_s pose_time_s poseest_time_s pose_estimator early_exit").split()


def save_per_sample_results(states, per_sample_file, pose_method, early_exit_status):
    """The per-sample CSV of utils/result_io.py:7-50: integers for the flag and the counts, 6 decimals for errors and times."""
    os.makedirs(os.path.dirname(per_sample_file), exist_ok=True)
    as_int, as_f6 = (0, 3, 4, 5, 6), (1, 2, 7, 8, 9, 10, 11)
    with open(per_sample_file, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(_CSV_HEAD)
        for k, st in enumerate(states):
            cells = {c: int(st[c]) for c in as_int}
            cells.update({c: f"{st[c]:.6f}" for c in as_f6})
            w.writerow([k] + [cells[c] for c in range(12)] + [pose_method, early_exit_status])


def evaluate_3dmatch(gtpath, benchmark, timestr, root="."):
    """test.py:280-306: RMSE recall per scene from the logs written by write_3dmatch_logs."""
    scenes = sorted(os.listdir(gtpath))
    rmse_recall = []
    for scene in scenes:
        gt_pairs, gt_traj = read_trajectory(os.path.join(gtpath, scene, "gt.log"))
        n_fragments, gt_traj_cov = read_trajectory_info(os.path.join(gtpath, scene, "gt.info"))
        est_pairs, est_traj = read_trajectory(os.path.join(root, f"logs/log_{benchmark}", scene, f"{timestr}.log"))
        _, rec, _, _ = evaluate_registration(n_fragments, est_traj, est_pairs, gt_pairs, gt_traj, gt_traj_cov)
        rmse_recall.append(rec)
    return scenes, rmse_recall
