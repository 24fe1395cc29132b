"""Evaluation row: buffer-x_amd/evaluate.py against tests/golden/eval/, which was minted by running the REAL
reference code (test.py's log-writer and statistics blocks, utils/tools.py, utils/SE3.py, utils/result_io.py; see
tests/golden/make_golden_eval.py).  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "eval")
TS = "20240101_000000"


@pytest.fixture(scope="module")
def ev():
    from bufferx_amd import evaluate
    return evaluate


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "eval.npz"), allow_pickle=False)


def _entries(gold):
    out = []
    for s, t, none, dt, e in zip(gold["src_ids"], gold["tgt_ids"], gold["none"], gold["dtypes"], gold["est64"]):
        out.append((str(s), str(t), None if none else e.astype(np.float32 if str(dt) == "float32" else np.float64)))
    return out


def test_log_writer_bytes_identical(tmp_path, ev, gold):
    paths = ev.write_3dmatch_logs("3DMatch", TS, _entries(gold), root=str(tmp_path))
    assert sorted(paths) == sorted(os.listdir(os.path.join(G, "logs", "log_3DMatch")))
    for scene, p in paths.items():
        ref = open(os.path.join(G, "logs", "log_3DMatch", scene, TS + ".log"), "rb").read()
        assert open(p, "rb").read() == ref


def test_trajectory_readers_and_rmse_recall(ev, gold):
    scenes, recalls = ev.evaluate_3dmatch(os.path.join(G, "gt_result"), "3DMatch", TS, root=G)
    for scene, rec in zip(scenes, recalls):
        gp, gt = ev.read_trajectory(os.path.join(G, "gt_result", scene, "gt.log"))
        nfr, cov = ev.read_trajectory_info(os.path.join(G, "gt_result", scene, "gt.info"))
        ep, et = ev.read_trajectory(os.path.join(G, "logs", "log_3DMatch", scene, TS + ".log"))
        assert np.array_equal(gp, gold[f"{scene}|gt_pairs"]) and np.array_equal(gt, gold[f"{scene}|gt_traj"])
        assert nfr == int(gold[f"{scene}|nfr"]) and np.array_equal(cov, gold[f"{scene}|gt_cov"])
        assert np.array_equal(ep, gold[f"{scene}|est_pairs"]) and np.array_equal(et, gold[f"{scene}|est_traj"])
        gl = ev.loadlog(os.path.join(G, "gt_result", scene))
        assert list(gl.keys()) == [str(k) for k in gold[f"{scene}|loadlog_keys"]]
        assert np.array_equal(np.array([gl[k] for k in gl]), gold[f"{scene}|loadlog_mats"])
        prec, rec2, flags, errs = ev.evaluate_registration(nfr, et, ep, gp, gt, cov)
        assert rec == rec2 == float(gold[f"{scene}|rec"]) and prec == float(gold[f"{scene}|prec"])
        assert np.array_equal(np.array(flags), gold[f"{scene}|flags"])
        ref = gold[f"{scene}|errs"]
        assert np.array_equal(np.isnan(errs), np.isnan(ref))
        m = ~np.isnan(ref)
        assert np.allclose(errs[m], ref[m], rtol=1e-6, atol=1e-12)     # mat2quat: eigen-solver here, scipy in the fixture


def test_rte_rre_states_summary_csv(tmp_path, ev, gold):
    states = gold["states"]
    for k, (relt, e) in enumerate(zip(gold["relt"], gold["est64"])):
        est = e.astype(np.float32) if str(gold["dtypes"][k]) == "float32" else e
        assert ev.compute_rte(est, relt) == states[k, 1] and ev.compute_rre(est, relt) == states[k, 2]
    s = ev.summarize(states)
    for k, v in zip(gold["summary_keys"], gold["summary_vals"]):
        assert s[str(k)] == v, k
    assert s["average_times"].shape == (5,) and np.array_equal(s["average_times"], states[5:, 7:12].mean(axis=0))
    f = str(tmp_path / "x" / "per_sample.csv")
    ev.save_per_sample_results(states, f, "RANSAC", "ON")
    assert open(f, "rb").read() == open(os.path.join(G, "per_sample.csv"), "rb").read()


def test_mat2quat_against_scipy(ev):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    mats = list(Rotation.random(200, random_state=1).as_matrix())
    mats += [np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.eye(3)]        # 180 degree turns, identity
    for M in mats:
        q = ev.mat2quat(M)
        x, y, z, w = Rotation.from_matrix(M).as_quat()
        r = np.array([w, x, y, z])
        assert q[0] >= 0 and abs(np.linalg.norm(q) - 1) < 1e-12
        assert min(np.abs(q - r).max(), np.abs(q + r).max()) < 1e-9


def test_pack_state_round_trip(ev, gold):
    rows = []
    for k, (relt, e) in enumerate(zip(gold["relt"], gold["est64"])):
        est = e.astype(np.float32) if str(gold["dtypes"][k]) == "float32" else e
        st = gold["states"][k]
        rows.append(ev.pack_state(k, est, relt, st[3], st[4], st[5], st[6], st[7], st[8], st[9:12], 0.3, 15))
    rows = ev.gather_states(np.stack(rows)[::-1], len(rows))                  # arrives in any order, leaves ordered by id
    assert np.array_equal(ev.states_matrix(rows), gold["states"])
    for k, r in enumerate(rows):
        p = ev.state_pose(r)
        assert str(p.dtype) == str(gold["dtypes"][k]) or bool(gold["none"][k])


WORKER = r'''
import os, sys, numpy as np, torch.distributed as dist
sys.path.insert(0, os.environ["BX_ROOT"])
from bufferx_amd import evaluate as ev, dist as D
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = np.load(os.path.join(os.environ["BX_GOLD"], "eval.npz"))
n = len(g["states"])
rows = []
for k in D.shard_indices(n, rank, world):
    e = g["est64"][k]
    est = None if g["none"][k] else (e.astype(np.float32) if str(g["dtypes"][k]) == "float32" else e)
    st = g["states"][k]
    rows.append(ev.pack_state(k, est, g["relt"][k], st[3], st[4], st[5], st[6], st[7], st[8], st[9:12], 0.3, 15))
allr = ev.gather_states(np.stack(rows), n)
assert np.array_equal(ev.states_matrix(allr), g["states"])
if rank == 0:
    entries = [(str(g["src_ids"][k]), str(g["tgt_ids"][k]), None if g["none"][k] else ev.state_pose(allr[k])) for k in range(n)]
    paths = ev.write_3dmatch_logs("3DMatch", "T", entries, root=os.environ["BX_OUT"])
    for scene, p in paths.items():
        ref = open(os.path.join(os.environ["BX_GOLD"], "logs", "log_3DMatch", scene, "20240101_000000.log"), "rb").read()
        assert open(p, "rb").read() == ref, scene
sys.stdout.write(f"rank{rank}ok\n"); sys.stdout.flush()
dist.destroy_process_group()
'''


def test_sharded_states_one_allgather_world2(tmp_path):
    """Pairs sharded round-robin over 2 ranks, ONE all-gather of float64 state rows, rank 0 writes the logs: same bytes."""
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, BX_ROOT=ROOT, BX_GOLD=G, BX_OUT=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(w)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank0ok" in out.stdout and "rank1ok" in out.stdout


def test_threedmatch_test_pair_list(tmp_path):
    """harness.threedmatch_test_pairs against the file list and poses of the real ThreeDMatchDataset(split="test")
    (tests/golden/make_golden_pairs.py)."""
    from bufferx_amd import harness
    g = np.load(os.path.join(G, "pairs.npz"))
    root = tmp_path / "data"
    (root / "test" / "3DMatch").mkdir(parents=True)
    os.symlink(os.path.join(G, "gt_result"), root / "test" / "3DMatch" / "gt_result")
    scenes = [s for s in harness.THREEDMATCH_TEST_SCENES if os.path.isdir(os.path.join(G, "gt_result", s))]
    pairs = harness.threedmatch_test_pairs(str(root), "3DMatch", scenes)
    assert [[p["src_id"], p["tgt_id"]] for p in pairs] == [list(map(str, f)) for f in g["files"]]
    for p, gt in zip(pairs, g["poses"]):
        assert np.array_equal(p["relt_pose"], np.linalg.inv(gt))            # dataset/threedmatch.py:123
        assert p["src_path"] == os.path.join(str(root), "test", p["src_id"]) + ".ply"
