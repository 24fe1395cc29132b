"""The reference's test loop (test.py:120-200) as a pipeline on one GPU, built from the rows of SURVEY.md §8:

    files --Prefetcher--> raw clouds in HBM --Preprocessor--> voxel size, first down-sampling, shuffle
          --bx_register_pair (C pairs in flight)--> pose --evaluate.pack_state--> one float64 row per pair

The reference does all of this serially per pair on the host thread (dataset/threedmatch.py:66-160 -> collate -> model -> metrics).
Here the files of the next pairs are parsed and uploaded by a native thread; the per-pair preparation (voxel analysis, down-sampling,
shuffle, the reference's NumPy draws, uploads) runs in a preparation thread on its own HIP stream and library context, several
pairs ahead; the registration thread only enqueues bx_register_pair for up to `inflight` pairs and harvests their results.

NumPy's global RNG is consumed by the same calls, in the same order, as dataset/threedmatch.py + models/patch_embedder.py make
them (analysis subsamples, shuffle of both clouds, the two shuffles of the second down-sampling, the per-scale permutations), so
a seeded run replays the reference's random choices; the RANSAC seed is one extra draw (Open3D's RANSAC is unseeded upstream)."""
import collections
import copy
import os
import queue
import threading
import time

import numpy as np

# the loop drives C registration streams + a preparation stream + the prefetcher's copy stream: with HIP's default of 4 hardware
# queues two of them would share a queue and the preparation would wait behind a whole pair (read at HIP initialisation)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import evaluate, ingest, lib
from .preprocess import Preprocessor


class Runner:
    def __init__(self, cfg, packed_weights, device=0, inflight=3, max_raw_points=400000, max_points=80000, rng="reference"):
        """rng: "reference" replays the loaders' legacy-NumPy draws call by call (about 20 ms of host time per pair); "device"
        computes every subsample / shuffle / permutation on the GPU (bx_random_perm, seeded from ONE np.random draw per pair)."""
        import torch
        self.torch, self.cfg, self.device = torch, cfg, int(device)
        assert rng in ("reference", "device")
        self.rng = rng
        self.C = max(1, int(inflight))
        self.max_points = int(max_points)
        self.ctxs = [lib.Context(cfg, max_points=self.max_points, device=self.device, packed_weights=packed_weights) for _ in range(self.C)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.C)]
        self.results = [c.new_result() for c in self.ctxs]
        # high priority: the preparation kernels are tiny and the host waits for three of their results per pair; they must not
        # queue behind the convolution kernels of the pairs in flight
        self.prep_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('BX_PREP_PRIO', '-1')))
        self._pin, self._pin_ev, self._pin_turn = {}, {}, {}
        self.timers = {k: 0.0 for k in ("wait_prefetch", "voxel_analysis", "down_sample", "shuffle", "second_sampling_rng", "perm_rng",
                                        "perm_upload", "wait_prepared", "harvest_wait", "enqueue")}   # host seconds, accumulated over run()
        # the preparation thread drives its OWN library context (a bx_ctx is not shared between threads): a minimal configuration,
        # it only serves bx_pre_* / bx_permute / bx_random_perm
        pc = copy.deepcopy(cfg)
        pc.patch.num_fps, pc.patch.num_points_per_patch, pc.patch.num_scales = 16, 16, 1
        pc.patch.search_radius_thresholds = [5]
        pc.patch.num_points_radius_estimate = 16
        pc.test.keypoint_tiles = 0
        self.prep_ctx = lib.Context(pc, max_points=self.max_points, device=self.device)
        self.pre = Preprocessor(self.prep_ctx, max_raw_points, upload=self._upload)   # bx_pre_* has its own workspace in the context
        # every device buffer of the loop is allocated ONCE: an allocation under load costs a hipMalloc, i.e. a device-wide wait.
        dev = f"cuda:{self.device}"
        S = int(cfg.patch.num_scales)
        f32, i32 = torch.float32, torch.int32
        self.sets = [dict(src=torch.empty((self.max_points, 3), dtype=f32, device=dev), tgt=torch.empty((self.max_points, 3), dtype=f32, device=dev),
                          perm_s=torch.empty(S * self.max_points, dtype=i32, device=dev), perm_t=torch.empty(S * self.max_points, dtype=i32, device=dev))
                     for _ in range(2 * self.C + 2)]     # pair i uses set i % (2C + 2): C pairs in flight + C + 2 being prepared
        self.scratch = dict(fds_s=torch.empty((max_raw_points, 3), dtype=f32, device=dev), fds_t=torch.empty((max_raw_points, 3), dtype=f32, device=dev),
                            sds_s=torch.empty((self.max_points, 3), dtype=f32, device=dev), sds_t=torch.empty((self.max_points, 3), dtype=f32, device=dev),
                            idx_s=torch.empty(max_raw_points, dtype=i32, device=dev), idx_t=torch.empty(max_raw_points, dtype=i32, device=dev))
        self._devbuf = {}
        self.pf = ingest.Prefetcher(device=self.device, slots=self.C + 2, max_points=max_raw_points)

    def close(self):
        self.pf.close()
        self.prep_ctx.close()
        for c in self.ctxs:
            c.close()

    def _upload(self, key, arr, into=None):
        """host array -> device through a reusable pinned staging buffer (a pageable source makes the copy synchronous)"""
        t = self.torch
        arr = np.ascontiguousarray(arr)
        # two staging buffers per key, used alternately: the wait below is for the copy issued two uploads ago, not for the one that
        # may still sit behind this pair's preparation kernels on the stream
        self._pin_turn[key] = turn = 1 - self._pin_turn.get(key, 1)
        pkey, key_ev = (key, turn), (key, turn)
        buf = self._pin.get(pkey)
        if buf is None or buf.numel() < arr.size or buf.dtype != t.from_numpy(arr).dtype:
            buf = t.empty(max(arr.size, 1), dtype=t.from_numpy(arr).dtype).pin_memory()
            self._pin[pkey] = buf
        else:
            self._pin_ev[key_ev].synchronize()           # the previous copy out of this buffer has completed
        view = buf[:arr.size].view(arr.shape) if arr.size else buf[:0]
        view.copy_(t.from_numpy(arr))
        dbuf = self._devbuf.get(key) if into is None else into
        if dbuf is None or dbuf.numel() < arr.size:
            dbuf = t.empty(max(arr.size, 1) * 2, dtype=buf.dtype, device=f"cuda:{self.device}")   # grows geometrically, rarely
            self._devbuf[key] = dbuf
        dev = dbuf[:arr.size].view(arr.shape) if arr.size else dbuf[:0]
        dev.copy_(view, non_blocking=True)
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self.device))
        self._pin_ev[key_ev] = ev
        return dev

    # ------------------------------------------------------------------------------------------ per-pair preparation
    def _prepare(self, ticket, voxel_size, replay_rng, bufs):
        """dataset/threedmatch.py:75-135 for the test split, on the GPU.  Returns (src, tgt, aligned_z, voxel_size, sphericity)."""
        cfg, pre, t = self.cfg, self.pre, self.torch
        tm = self.timers
        t0 = time.perf_counter()
        src_raw, tgt_raw = self.pf.wait(ticket)
        tm["wait_prefetch"] += time.perf_counter() - t0; t0 = time.perf_counter()
        sphericity = 0.0
        sc = self.scratch
        c0 = self.prep_ctx
        dev_rng = self.rng == "device"
        if dev_rng:
            base = int(np.random.randint(0, 2**31 - 1)) << 8          # the ONE host draw of this pair; +k = its k-th device stream
        if voxel_size is None:
            sidx = None
            if dev_rng:     # a 10 % subsample without replacement = the head of a random permutation
                ns_, nt_ = src_raw.shape[0], tgt_raw.shape[0]
                sidx = (c0.random_perm(ns_, base + 0, out=sc["idx_s"])[:ns_ // 10], c0.random_perm(nt_, base + 1, out=sc["idx_t"])[:nt_ // 10])
            voxel_size, sphericity, _ = pre.sphericity_based_voxel_analysis(src_raw, tgt_raw, sample_idx=sidx)
        tm["voxel_analysis"] += time.perf_counter() - t0; t0 = time.perf_counter()
        src, tgt = pre.voxel_down_sample_many([src_raw, tgt_raw], voxel_size, outs=[sc["fds_s"], sc["fds_t"]])
        self.pf.release(ticket)
        tm["down_sample"] += time.perf_counter() - t0; t0 = time.perf_counter()
        if src.shape[0] > self.max_points or tgt.shape[0] > self.max_points:
            raise lib.BxError(f"down-sampled cloud of {max(src.shape[0], tgt.shape[0])} points exceeds max_points={self.max_points}")
        # np.random.shuffle(pts) == pts[np.random.permutation(len(pts))], same RNG consumption
        if dev_rng:
            src = c0.permute(src, c0.random_perm(src.shape[0], base + 2, out=sc["idx_s"]), out=bufs["src"])
            tgt = c0.permute(tgt, c0.random_perm(tgt.shape[0], base + 3, out=sc["idx_t"]), out=bufs["tgt"])
            S = int(cfg.patch.num_scales)
            ns_, nt_ = src.shape[0], tgt.shape[0]
            for k in range(S):      # per-scale permutations straight into this pair's buffers
                c0.random_perm(ns_, base + 4 + 2 * k, out=bufs["perm_s"][k * ns_:(k + 1) * ns_])
                c0.random_perm(nt_, base + 5 + 2 * k, out=bufs["perm_t"][k * nt_:(k + 1) * nt_])
            self._dev_perms = (bufs["perm_s"][:S * ns_].view(S, ns_), bufs["perm_t"][:S * nt_].view(S, nt_), base + 255)
            tm["shuffle"] += time.perf_counter() - t0
            return src, tgt, bool(cfg.patch.is_aligned_to_global_z), voxel_size, sphericity
        src = c0.permute(src, self._upload("shuf_s", np.random.permutation(src.shape[0]).astype(np.int32)), out=bufs["src"])
        tgt = c0.permute(tgt, self._upload("shuf_t", np.random.permutation(tgt.shape[0]).astype(np.int32)), out=bufs["tgt"])
        tm["shuffle"] += time.perf_counter() - t0; t0 = time.perf_counter()
        if replay_rng:
            # the loader's second down-sampling only feeds training, but its shuffles (and the max_numPts subsample) advance the RNG
            vs0 = float(cfg.data.voxel_size_0)
            max_n = int(cfg.data.get("max_numPts", 30000))
            counts = [int(x.shape[0]) for x in pre.voxel_down_sample_many([src, tgt], vs0, outs=[sc["sds_s"], sc["sds_t"]])]
            for m in counts:
                np.random.permutation(m)
            for m in counts:
                if m > max_n:
                    np.random.choice(range(m), max_n, replace=False)
        tm["second_sampling_rng"] += time.perf_counter() - t0
        return src, tgt, bool(cfg.patch.is_aligned_to_global_z), voxel_size, sphericity

    # ------------------------------------------------------------------------------------------ the loop
    def run(self, pairs, voxel_size=None, replay_rng=True, rank=0, world=1, pair_seed=None):
        """pairs: list of dicts {src_path, tgt_path, relt_pose [4,4]} (+ anything else, carried through).
        -> (rows float64 [n_mine, evaluate.STATE_W] ordered by pair index, poses list indexed like `pairs`, None for pairs of other
        ranks); rows feed evaluate.gather_states / summarize.

        Sharding (the reference's loop, test.py:132-146, is one process): rank r of `world` takes the pairs {i : i mod world == r}
        (dist.shard_indices); row ids are GLOBAL pair indices, so evaluate.gather_states() of every rank's rows is the single-rank
        result.  The reference draws every random number of the loop from ONE global NumPy stream, which no sharded run can replay;
        `pair_seed` makes a pair's draws a function of the pair alone (np.random.seed(pair_seed + i) in front of pair i), and then
        any world size produces the same rows bit for bit.  Without it, rng="device" still shards reproducibly (one host draw per
        pair, consumed for the pairs of the other ranks too); rng="reference" refuses to shard."""
        t, cfg, C = self.torch, self.cfg, self.C
        S = int(cfg.patch.num_scales)
        if world > 1 and pair_seed is None and self.rng == "reference":
            raise ValueError("rng='reference' replays ONE global NumPy stream: a sharded run needs pair_seed (per-pair streams)")
        n_all = len(pairs)
        mine = list(range(rank, n_all, world))
        gids = mine                                   # local position -> global pair index
        pairs = [pairs[i] for i in mine]
        n = len(pairs)
        rows, poses = [None] * n, [None] * n_all
        pending = [None] * C
        NSET = len(self.sets)
        permits = threading.Semaphore(NSET)           # buffer set i % NSET is free again once pair i - NSET has been harvested
        ready_q = queue.Queue()

        # ---- the preparation thread: everything of a pair that happens on the host before bx_register_pair -- wait for the prefetched
        # files, voxel analysis, down-sampling, shuffle, the reference's np.random draws (ONE thread makes all of them, in the
        # reference's order) and the uploads -- runs up to NSET pairs ahead of the pairs in flight, on its own stream and its own
        # library context.  The registration thread below only enqueues bx_register_pair and harvests results.
        def producer():
            try:
                depth = min(n, C + 2)
                tickets = collections.deque(self.pf.submit(p["src_path"], p["tgt_path"]) for p in pairs[:depth])
                submitted = depth
                next_draw = 0                         # global index of the next pair whose host draw has not been consumed
                for i in range(n):
                    permits.acquire()
                    if self._stop:
                        return
                    t0 = time.perf_counter()
                    if pair_seed is not None:
                        np.random.seed((int(pair_seed) + gids[i]) % (2 ** 32))
                    elif world > 1:
                        # rng="device": the pairs of the other ranks consume their ONE host draw too (same stream as a single rank)
                        while next_draw < gids[i]:
                            np.random.randint(0, 2**31 - 1)
                            next_draw += 1
                        next_draw = gids[i] + 1
                    with t.cuda.stream(self.prep_stream):
                        bufs = self.sets[i % NSET]
                        src, tgt, aligned, _, _ = self._prepare(tickets.popleft(), voxel_size, replay_rng, bufs)
                        tq = time.perf_counter()
                        if self.rng == "device":
                            d_ps, d_pt, seed = self._dev_perms
                        else:
                            perm_s, perm_t = [], []
                            for _ in range(S):        # models/patch_embedder.py:96, order scale0-src, scale0-tgt, scale1-src, ...
                                perm_s.append(np.random.choice(src.shape[0], src.shape[0], replace=False).astype(np.int32))
                                perm_t.append(np.random.choice(tgt.shape[0], tgt.shape[0], replace=False).astype(np.int32))
                            seed = int(np.random.randint(0, 2**31 - 1))
                            self.timers["perm_rng"] += time.perf_counter() - tq; tq = time.perf_counter()
                            d_ps = self._upload("perm_s", np.stack(perm_s), into=bufs["perm_s"])
                            d_pt = self._upload("perm_t", np.stack(perm_t), into=bufs["perm_t"])
                            self.timers["perm_upload"] += time.perf_counter() - tq
                        ready = t.cuda.Event()
                        ready.record(self.prep_stream)
                    if submitted < n:
                        tickets.append(self.pf.submit(pairs[submitted]["src_path"], pairs[submitted]["tgt_path"]))
                        submitted += 1
                    ready_q.put((src, tgt, aligned, d_ps, d_pt, seed, ready, time.perf_counter() - t0))
            except BaseException as e:                # surfaces in the registration thread
                ready_q.put(e)

        def harvest(c):
            if pending[c] is None:
                return
            i, a, b, data_s = pending[c]
            self.streams[c].synchronize()
            r = self.results[c]
            if r.status != 0:
                raise lib.BxError(f"pair {i}: device-side failure bits 0x{r.status:x}")
            pose = np.array(r.pose, np.float64).reshape(4, 4)
            if cfg.test.pose_refine is True:
                pose = pose.astype(np.float32)
            poses[gids[i]] = pose
            # the reference's collate hands the ground truth over as float32 (dataset/dataloader.py:113), so RTE / RRE of a refined
            # (float32) pose are float32 arithmetic in test.py:168-170: same dtype here, or the 6-decimal CSV cells can differ
            gt = np.asarray(pairs[i]["relt_pose"], np.float32)
            rows[i] = evaluate.pack_state(gids[i], pose, gt, r.num_inliers, r.num_mutual, r.num_inlier_ind,
                                          r.scales_used, data_s, a.elapsed_time(b) / 1e3, [0.0, 0.0, 0.0],
                                          cfg.test.rte_thresh, cfg.test.rre_thresh)
            pending[c] = None
            permits.release()

        self._stop = False
        th = threading.Thread(target=producer, name="bx-prepare", daemon=True)
        th.start()
        try:
            for i in range(n):
                tq = time.perf_counter()
                item = ready_q.get()
                if isinstance(item, BaseException):
                    raise item
                src, tgt, aligned, d_ps, d_pt, seed, ready, data_s = item
                self.timers["wait_prepared"] += time.perf_counter() - tq; tq = time.perf_counter()
                c = i % C
                harvest(c)
                self.timers["harvest_wait"] += time.perf_counter() - tq; tq = time.perf_counter()
                st = self.streams[c]
                st.wait_event(ready)
                with t.cuda.stream(st):
                    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                    a.record(st)
                    self.ctxs[c].register_pair_async(src, tgt, aligned, d_ps, d_pt, seed, self.results[c])
                    b.record(st)
                pending[c] = (i, a, b, data_s)
                self.timers["enqueue"] += time.perf_counter() - tq
            for c in range(C):
                harvest(c)
        finally:
            self._stop = True
            for _ in range(NSET + 1):                 # a producer parked on the semaphore wakes up and leaves
                permits.release()
            th.join(timeout=60)
        return np.stack(rows) if n else np.zeros((0, evaluate.STATE_W)), poses


# ---------------------------------------------------------------------------------------------------- 3DMatch test split
THREEDMATCH_TEST_SCENES = ("7-scenes-redkitchen", "sun3d-home_at-home_at_scan1_2013_jan_1", "sun3d-home_md-home_md_scan9_2012_sep_30",
                           "sun3d-hotel_uc-scan3", "sun3d-hotel_umd-maryland_hotel1", "sun3d-hotel_umd-maryland_hotel3",
                           "sun3d-mit_76_studyroom-76-1studyroom2", "sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika")


def threedmatch_test_pairs(root, benchmark="3DMatch", scenes=THREEDMATCH_TEST_SCENES):
    """The pair list of ThreeDMatchDataset(split="test") (dataset/threedmatch.py:36-63) in the form Runner.run takes: one dict per
    gt.log entry, in file order, with src_id / tgt_id as the loader names them ("3DMatch/fragments/<scene>/cloud_bin_<k>"), the
    .ply paths under <root>/test (dataset/threedmatch.py:73-79) and relt_pose = inv(gt) (dataset/threedmatch.py:123)."""
    test_root = os.path.join(root, "test")
    pairs = []
    for scene in scenes:
        gtpath = test_root + (f"/{benchmark}/gt_result/{scene}" if benchmark == "3DMatch" else f"/{benchmark}/{scene}")
        for key, gt in evaluate.loadlog(gtpath).items():
            id1, id2 = key.split("_")[0], key.split("_")[1]
            src_id = os.path.join(f"3DMatch/fragments/{scene}", f"cloud_bin_{id1}")
            tgt_id = os.path.join(f"3DMatch/fragments/{scene}", f"cloud_bin_{id2}")
            pairs.append(dict(src_id=src_id, tgt_id=tgt_id, src_path=os.path.join(test_root, src_id) + ".ply",
                              tgt_path=os.path.join(test_root, tgt_id) + ".ply", relt_pose=np.linalg.inv(gt)))
    return pairs


def run_3dmatch(cfg, packed_weights, root, benchmark="3DMatch", timestr="run", out_root=".", scenes=THREEDMATCH_TEST_SCENES,
                rank=0, world=1, pair_seed=None, collective_device=None, **runner_kw):
    """test.py for the 3DMatch / 3DLoMatch test split: pair list -> Runner -> .log files -> RMSE recall + summary.
    world > 1 (one process per GPU, torch.distributed initialised by the caller): every rank registers its pairs {i mod world ==
    rank}, ONE all-gather of the float64 state rows (evaluate.gather_states; RCCL when collective_device is a cuda device) makes
    every rank hold all rows, rank 0 writes the .log files and evaluates them; the other ranks return (rows, None)."""
    pairs = threedmatch_test_pairs(root, benchmark, scenes)
    run = Runner(cfg, packed_weights, **runner_kw)
    try:
        rows, poses = run.run(pairs, rank=rank, world=world, pair_seed=pair_seed)
    finally:
        run.close()
    if world > 1:
        rows = evaluate.gather_states(rows, len(pairs), device=collective_device)
        poses = [evaluate.state_pose(r) for r in rows]
        if rank != 0:
            return rows, None
    evaluate.write_3dmatch_logs(benchmark, timestr, [(p["src_id"], p["tgt_id"], pose) for p, pose in zip(pairs, poses)], root=out_root)
    gtpath = os.path.join(root, "test", benchmark, "gt_result") if benchmark == "3DMatch" else os.path.join(root, "test", benchmark)
    scenes, rmse_recall = evaluate.evaluate_3dmatch(gtpath, benchmark, timestr, root=out_root)
    summary = evaluate.summarize(evaluate.states_matrix(rows))
    summary["rmse_recall"] = float(np.mean(rmse_recall))
    summary["scene_recall"] = dict(zip(scenes, rmse_recall))
    return rows, summary
