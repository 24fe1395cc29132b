"""End-to-end pipeline on the GPU (buffer-x_amd/harness.py): files -> prefetch -> GPU pre-processing -> pairs in flight -> metrics,
against the same steps made one by one, synchronously, with the same NumPy seed (bit-identical poses), and the evaluation rows."""
import os

import numpy as np
import pytest

from oracle import io_oracle as IO

pytestmark = pytest.mark.gpu


def _cfg(bx):
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 256, 128, 2
    cfg.patch.search_radius_thresholds = [5, 2]
    cfg.patch.num_points_radius_estimate = 256
    cfg.test.pose_refine = True
    return cfg


def test_runner_matches_step_by_step(tmp_path, bx, packed):
    import torch
    from bufferx_amd import evaluate, harness, ingest, lib
    from bufferx_amd.preprocess import Preprocessor
    rng = np.random.default_rng(3)
    pairs = []
    for i in range(4):
        p = bx.synth.make_pair(20 + i, "indoor", n_target=9000, jitter=0.0)
        # raw scans: every surface point several times with millimetre jitter, so that the first down-sampling has work to do
        raw = [np.concatenate([c + rng.normal(0, 0.002, c.shape) for _ in range(3)]).astype(np.float32) for c in (p["src"], p["tgt"])]
        fs, ft = str(tmp_path / f"s{i}.ply"), str(tmp_path / (f"t{i}.pcd" if i % 2 else f"t{i}.ply"))
        IO.write_ply(fs, raw[0])
        (IO.write_pcd(ft, raw[1], "binary_compressed") if i % 2 else IO.write_ply(ft, raw[1], "binary_big_endian"))
        pairs.append(dict(src_path=fs, tgt_path=ft, relt_pose=p["T_gt"]))
    cfg = _cfg(bx)

    np.random.seed(11)
    run = harness.Runner(cfg, packed, device=0, inflight=3, max_raw_points=40000, max_points=40000)
    try:
        rows, poses = run.run(pairs)
    finally:
        run.close()
    assert rows.shape == (4, evaluate.STATE_W) and list(rows[:, 0]) == [0, 1, 2, 3]

    # the same steps one by one
    np.random.seed(11)
    ctx = lib.Context(cfg, max_points=40000, device=0, packed_weights=packed)
    pre = Preprocessor(ctx, 40000)
    try:
        for i, p in enumerate(pairs):
            src_raw, tgt_raw = ingest.read_point_cloud(p["src_path"]), ingest.read_point_cloud(p["tgt_path"])
            vs, _, _ = pre.sphericity_based_voxel_analysis(src_raw, tgt_raw)
            src, tgt = pre.voxel_down_sample(src_raw, vs), pre.voxel_down_sample(tgt_raw, vs)
            src = ctx.permute(src, np.random.permutation(src.shape[0]).astype(np.int32))
            tgt = ctx.permute(tgt, np.random.permutation(tgt.shape[0]).astype(np.int32))
            counts = [pre.voxel_down_sample(x, cfg.data.voxel_size_0).shape[0] for x in (src, tgt)]
            for m in counts:
                np.random.permutation(m)
            ps, pt = [], []
            for _ in range(2):
                ps.append(np.random.choice(src.shape[0], src.shape[0], replace=False).astype(np.int32))
                pt.append(np.random.choice(tgt.shape[0], tgt.shape[0], replace=False).astype(np.int32))
            seed = int(np.random.randint(0, 2**31 - 1))
            res = ctx.register_pair(src, tgt, cfg.patch.is_aligned_to_global_z, np.stack(ps), np.stack(pt), seed)
            pose = np.array(res.pose, np.float64).reshape(4, 4).astype(np.float32)
            assert np.array_equal(pose, poses[i]), i
            if i == 0:
                # ... and the pipeline is not only consistent with itself: the registration of the first pair, on the clouds the GPU
                # pre-processing produced and with the permutations / seed the loop drew, equals the CPU oracle's bit for bit
                from oracle import pipeline as PL
                ref = PL.register_pair(src.cpu().numpy(), tgt.cpu().numpy(), packed, cfg, cfg.patch.is_aligned_to_global_z, seed,
                                       perms=(np.stack(ps), np.stack(pt)))
                assert (res.num_inliers, res.num_mutual, res.num_inlier_ind, res.scales_used) == tuple(ref[1:])
                assert np.array_equal(np.array(res.pose, np.float64).reshape(4, 4), np.asarray(ref[0], np.float64))
            assert rows[i, 4] == res.num_inliers and rows[i, 5] == res.num_mutual and rows[i, 7] == res.scales_used
            assert rows[i, 2] == evaluate.compute_rte(pose, np.asarray(p["relt_pose"], np.float32))      # float32 ground truth like the reference collate
            assert np.array_equal(evaluate.state_pose(rows[i]), pose)
    finally:
        ctx.close()
    s = evaluate.summarize(evaluate.states_matrix(rows))
    assert 0.0 <= s["recall"] <= 1.0 and s["average_times"].shape == (5,)


def test_runner_device_rng_is_deterministic(tmp_path, bx, packed):
    """rng="device": subsamples, shuffles and per-scale permutations come from bx_random_perm; one NumPy draw per pair seeds them."""
    from bufferx_amd import evaluate, harness
    rng = np.random.default_rng(5)
    pairs = []
    for i in range(3):
        p = bx.synth.make_pair(40 + i, "indoor", n_target=8000, jitter=0.0)
        raw = [np.concatenate([c + rng.normal(0, 0.002, c.shape) for _ in range(3)]).astype(np.float32) for c in (p["src"], p["tgt"])]
        fs, ft = str(tmp_path / f"s{i}.ply"), str(tmp_path / f"t{i}.ply")
        IO.write_ply(fs, raw[0]); IO.write_ply(ft, raw[1])
        pairs.append(dict(src_path=fs, tgt_path=ft, relt_pose=p["T_gt"]))
    out = []
    for _ in range(2):
        np.random.seed(3)
        run = harness.Runner(_cfg(bx), packed, device=0, inflight=2, max_raw_points=40000, max_points=40000, rng="device")
        try:
            out.append(run.run(pairs))
        finally:
            run.close()
    assert np.array_equal(out[0][0][:, :8], out[1][0][:, :8]) and np.array_equal(out[0][0][:, 13:], out[1][0][:, 13:])
    assert all(np.array_equal(a, b) for a, b in zip(out[0][1], out[1][1]))
    assert (out[0][0][:, 5] > 0).all()        # mutual matches were found: the permutations fed real neighbourhoods


def test_run_3dmatch_end_to_end(tmp_path, bx, packed):
    """A miniature 3DMatch test split on disk (fragments as .ply, gt.log, gt.info) through harness.run_3dmatch: pair list -> pipeline ->
    .log files -> RMSE recall; noise-free pairs with the real thresholds must register."""
    from bufferx_amd import evaluate, harness
    scene = harness.THREEDMATCH_TEST_SCENES[0]
    root = tmp_path / "data"
    frag = root / "test" / "3DMatch" / "fragments" / scene
    gtd = root / "test" / "3DMatch" / "gt_result" / scene
    frag.mkdir(parents=True); gtd.mkdir(parents=True)
    log, info = [], []
    for k, (i, j) in enumerate([(0, 2), (3, 5)]):
        p = bx.synth.make_pair(60 + k, "indoor", n_target=9000, jitter=0.0, identical=True)
        IO.write_ply(str(frag / f"cloud_bin_{i}.ply"), p["src"])
        IO.write_ply(str(frag / f"cloud_bin_{j}.ply"), p["tgt"])
        G = np.linalg.inv(p["T_gt"])                      # the loader uses relt_pose = inv(gt.log entry)
        log.append(f"{i}\t {j}\t 6\n" + "".join("\t".join(repr(float(v)) for v in G[r]) + "\t\n" for r in range(4)))
        info.append(f"{i}\t {j}\t 6\n" + "".join("\t".join(repr(float(v)) for v in np.eye(6)[r] * 100) + "\t\n" for r in range(6)))
    (gtd / "gt.log").write_text("".join(log)); (gtd / "gt.info").write_text("".join(info))
    cfg = _cfg(bx)
    np.random.seed(2)
    rows, summary = harness.run_3dmatch(cfg, packed, str(root), "3DMatch", "T0", out_root=str(tmp_path), scenes=[scene],
                                        inflight=2, max_raw_points=40000, max_points=40000, rng="device")
    assert rows.shape[0] == 2 and os.path.exists(tmp_path / "logs" / "log_3DMatch" / scene / "T0.log")
    assert set(summary["scene_recall"]) == {scene} and 0.0 <= summary["rmse_recall"] <= 1.0
    est_pairs, est_traj = evaluate.read_trajectory(str(tmp_path / "logs" / "log_3DMatch" / scene / "T0.log"))
    assert est_pairs[:, :2].tolist() == [["0", "2"], ["3", "5"]] and est_traj.shape == (2, 4, 4)
