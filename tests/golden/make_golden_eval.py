"""tests/golden/make_golden_eval.py -- mints tests/golden/eval/ for the evaluation row by running the REAL
reference code in this container:
  * the 3DMatch .log writer: the statement block of test.py (between `trans_est = trans_est if ...` and `####### Evaluation`) is
    read from /root/reference/test.py and exec'd unmodified, pair by pair, exactly as the test loop does;
  * utils/tools.py::read_trajectory / read_trajectory_info / evaluate_registration (nibabel is absent: nq.mat2quat is provided by
    scipy.spatial.transform.Rotation, an implementation independent of buffer-x_amd/evaluate.py::mat2quat);
  * utils/SE3.py::compute_rte / compute_rre, utils/result_io.py::save_per_sample_results and the summary statistics block of
    test.py (also exec'd from its source).
Run: python tests/golden/make_golden_eval.py   (needs /root/reference; the fixtures are committed)."""
import os
import shutil
import sys
import textwrap
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "eval")
REF = "/root/reference"


def rot(rng, deg):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def se3(R, t):
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T


def source_block(text, start_marker, end_marker, after=None):
    i0 = text.index(after) if after else 0
    a = text.index(start_marker, i0)
    a = text.rfind("\n", 0, a) + 1
    b = text.index(end_marker, a)
    b = text.rfind("\n", 0, b) + 1
    return textwrap.dedent(text[a:b])


def main():
    from scipy.spatial.transform import Rotation
    for name in ("open3d", "nibabel", "nibabel.quaternions"):
        sys.modules.setdefault(name, types.ModuleType(name))
    nq = sys.modules["nibabel.quaternions"]
    sys.modules["nibabel"].quaternions = nq

    def mat2quat(r):
        x, y, z, w = Rotation.from_matrix(np.asarray(r, np.float64)).as_quat()
        q = np.array([w, x, y, z])
        return -q if q[0] < 0 else q
    nq.mat2quat = mat2quat
    sys.path.insert(0, REF)
    from utils import tools as T
    from utils import SE3
    from utils import result_io
    test_src = open(os.path.join(REF, "test.py")).read()
    log_block = source_block(test_src, 'if cfg.data.dataset == "3DMatch":', "####### Evaluation #######", after="trans_est = trans_est if trans_est is not None")
    stats_block = "states = np.array(states)\n" + source_block(test_src, "recall = states[:, 0].sum()", "# Save per-sample results to csv file")

    if os.path.exists(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    rng = np.random.default_rng(2024)
    gtroot = os.path.join(OUT, "gt_result")
    cwd = os.getcwd()
    os.chdir(OUT)                               # the reference writes logs/ relative to the working directory
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(dataset="3DMatch", benchmark="3DMatch"))
    timestr = "20240101_000000"
    states, entries, gts = [], [], []
    pid = 0
    for scene, nfrag in (("7-scenes-redkitchen", 9), ("sun3d-hotel_umd-maryland_hotel3", 7)):
        os.makedirs(os.path.join(gtroot, scene))
        pairs = [(i, j) for i in range(nfrag) for j in range(i + 1, nfrag) if rng.random() < 0.6]
        with open(os.path.join(gtroot, scene, "gt.log"), "w") as fl, open(os.path.join(gtroot, scene, "gt.info"), "w") as fi:
            for (i, j) in pairs:
                Tg = se3(rot(rng, rng.uniform(0, 60)), rng.normal(size=3))
                fl.write(f"{i}\t {j}\t {nfrag}\n")
                for r in range(4):
                    fl.write("\t".join(f"{v:.8e}" for v in Tg[r]) + "\t\n")
                A = rng.normal(size=(6, 6)); info = A @ A.T * 1000 + np.eye(6) * 50
                fi.write(f"{i}\t {j}\t {nfrag}\n")
                for r in range(6):
                    fi.write("\t".join(f"{v:.8e}" for v in info[r]) + "\t\n")
                # the harness logs inv(trans_est): est = inv(gt) perturbed.  Small / large perturbations, f32 / f64 poses.
                kind = rng.integers(0, 4)
                dR = rot(rng, [0.3, 1.5, 8.0, 40.0][kind]); dt = rng.normal(size=3) * [0.005, 0.03, 0.15, 0.8][kind]
                est = np.linalg.inv(Tg) @ se3(dR, dt)
                est = est.astype(np.float32) if rng.random() < 0.5 else est
                if rng.random() < 0.05:
                    est = None
                data_source = {"src_id": f"3DMatch/test/{scene}/cloud_bin_{i}", "tgt_id": f"3DMatch/test/{scene}/cloud_bin_{j}"}
                trans_est = est if est is not None else np.eye(4)
                exec(log_block, {"cfg": cfg, "data_source": data_source, "trans_est": trans_est, "timestr": timestr, "os": os, "np": np})
                relt = np.linalg.inv(Tg)
                rte, rre = SE3.compute_rte(trans_est, relt), SE3.compute_rre(trans_est, relt)
                times = rng.random(3) * 0.05
                states.append([rte < 0.3 and rre < 15, rte, rre, rng.integers(0, 500), rng.integers(0, 900), rng.integers(0, 300),
                               rng.integers(1, 4), rng.random() * 0.01, rng.random() * 0.06, *times])
                entries.append((data_source["src_id"], data_source["tgt_id"], pid, est is None, None if est is None else str(est.dtype)))
                gts.append((relt, trans_est))
                pid += 1
    # evaluation with the real functions
    scenes = sorted(os.listdir(gtroot))
    ev = {}
    for scene in scenes:
        gt_pairs, gt_traj = T.read_trajectory(os.path.join(gtroot, scene, "gt.log"))
        nfr, gt_cov = T.read_trajectory_info(os.path.join(gtroot, scene, "gt.info"))
        est_pairs, est_traj = T.read_trajectory(os.path.join("logs/log_3DMatch", scene, f"{timestr}.log"))
        prec, rec, flags, errs = T.evaluate_registration(nfr, est_traj, est_pairs, gt_pairs, gt_traj, gt_cov)
        gl = T.loadlog(os.path.join(gtroot, scene))
        ev[scene] = dict(loadlog_keys=np.array(list(gl.keys())), loadlog_mats=np.array([gl[k] for k in gl]), prec=prec, rec=rec, flags=np.array(flags), errs=errs, nfr=nfr, gt_pairs=gt_pairs, gt_traj=gt_traj, gt_cov=gt_cov,
                         est_pairs=est_pairs, est_traj=est_traj)
    ns = {"np": np, "states": [list(map(float, s)) for s in states]}
    exec(stats_block, ns)
    result_io.save_per_sample_results(np.array(ns["states"]), os.path.join(OUT, "per_sample.csv"), "RANSAC", "ON")
    os.chdir(cwd)
    flat = {}
    for scene, d in ev.items():
        for k, v in d.items():
            flat[f"{scene}|{k}"] = v
    summary = {k: ns[k] for k in ("recall", "rte_mean", "rre_mean", "rte_std", "rre_std", "inliers_mean", "inliers_std", "mutual_inliers_mean",
                                  "mutual_inliers_std", "inlier_ind_mean", "inlier_ind_std", "scales_used_mean", "scales_used_std")}
    np.savez_compressed(os.path.join(OUT, "eval.npz"), states=np.array(ns["states"]),
                        src_ids=np.array([e[0] for e in entries]), tgt_ids=np.array([e[1] for e in entries]),
                        none=np.array([e[3] for e in entries]), dtypes=np.array([str(e[4]) for e in entries]),
                        relt=np.array([g[0] for g in gts]), est64=np.array([np.asarray(g[1], np.float64) for g in gts]),
                        summary_keys=np.array(list(summary)), summary_vals=np.array([summary[k] for k in summary]), **flat)
    print("pairs", pid, {s: (ev[s]["prec"], ev[s]["rec"]) for s in ev}, "recall", summary["recall"])


if __name__ == "__main__":
    main()
