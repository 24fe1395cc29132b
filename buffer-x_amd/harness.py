"""The reference's test loop (test.py:120-200) as a pipeline on one GPU, built from the rows of SURVEY.md §8:

    files --Prefetcher--> raw clouds in HBM --Preprocessor--> voxel size, first down-sampling, shuffle
          --bx_register_pair (C pairs in flight)--> pose --evaluate.pack_state--> one float64 row per pair

The reference does all of this serially per pair on the host thread (dataset/threedmatch.py:66-160 -> collate -> model -> metrics).
Here the files of the next pairs are parsed and uploaded by a native thread, the per-pair preparation runs on its own HIP stream
while up to `inflight` earlier pairs occupy the GPU, and nothing on the registration streams waits for the host.

NumPy's global RNG is consumed by the same calls, in the same order, as dataset/threedmatch.py + models/patch_embedder.py make
them (analysis subsamples, shuffle of both clouds, the two shuffles of the second down-sampling, the per-scale permutations), so
a seeded run replays the reference's random choices; the RANSAC seed is one extra draw (Open3D's RANSAC is unseeded upstream)."""
import time

import numpy as np

from . import evaluate, ingest, lib
from .preprocess import Preprocessor


class Runner:
    def __init__(self, cfg, packed_weights, device=0, inflight=3, max_raw_points=400000, max_points=80000):
        import torch
        self.torch, self.cfg, self.device = torch, cfg, int(device)
        self.C = max(1, int(inflight))
        self.max_points = int(max_points)
        self.ctxs = [lib.Context(cfg, max_points=self.max_points, device=self.device, packed_weights=packed_weights) for _ in range(self.C)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.C)]
        self.results = [c.new_result() for c in self.ctxs]
        self.prep_stream = torch.cuda.Stream(device=self.device)
        self.pre = Preprocessor(self.ctxs[0], max_raw_points)       # bx_pre_* has its own workspace inside the context
        self.pf = ingest.Prefetcher(device=self.device, slots=self.C + 2, max_points=max_raw_points)

    def close(self):
        self.pf.close()
        for c in self.ctxs:
            c.close()

    # ------------------------------------------------------------------------------------------ per-pair preparation
    def _prepare(self, ticket, voxel_size, replay_rng):
        """dataset/threedmatch.py:75-135 for the test split, on the GPU.  Returns (src, tgt, aligned_z, voxel_size, sphericity)."""
        cfg, pre, t = self.cfg, self.pre, self.torch
        src_raw, tgt_raw = self.pf.wait(ticket)
        sphericity = 0.0
        if voxel_size is None:
            voxel_size, sphericity, _ = pre.sphericity_based_voxel_analysis(src_raw, tgt_raw)
        src = pre.voxel_down_sample(src_raw, voxel_size)
        tgt = pre.voxel_down_sample(tgt_raw, voxel_size)
        self.pf.release(ticket)
        if src.shape[0] > self.max_points or tgt.shape[0] > self.max_points:
            raise lib.BxError(f"down-sampled cloud of {max(src.shape[0], tgt.shape[0])} points exceeds max_points={self.max_points}")
        # np.random.shuffle(pts) == pts[np.random.permutation(len(pts))], same RNG consumption
        ps = t.from_numpy(np.random.permutation(src.shape[0]).astype(np.int32))
        src = self.ctxs[0].permute(src, ps)
        pt = t.from_numpy(np.random.permutation(tgt.shape[0]).astype(np.int32))
        tgt = self.ctxs[0].permute(tgt, pt)
        if replay_rng:
            # the loader's second down-sampling only feeds training, but its shuffles (and the max_numPts subsample) advance the RNG
            vs0 = float(cfg.data.voxel_size_0)
            max_n = int(cfg.data.get("max_numPts", 30000))
            counts = [int(pre.voxel_down_sample(x, vs0).shape[0]) for x in (src, tgt)]
            for m in counts:
                np.random.permutation(m)
            for m in counts:
                if m > max_n:
                    np.random.choice(range(m), max_n, replace=False)
        return src, tgt, bool(cfg.patch.is_aligned_to_global_z), voxel_size, sphericity

    # ------------------------------------------------------------------------------------------ the loop
    def run(self, pairs, voxel_size=None, replay_rng=True):
        """pairs: list of dicts {src_path, tgt_path, relt_pose [4,4]} (+ anything else, carried through).
        -> (rows float64 [n, evaluate.STATE_W] ordered by pair index, poses list) ; rows feed evaluate.gather_states / summarize."""
        t, cfg, C = self.torch, self.cfg, self.C
        S = int(cfg.patch.num_scales)
        n = len(pairs)
        depth = min(n, C + 2)
        tickets = [self.pf.submit(p["src_path"], p["tgt_path"]) for p in pairs[:depth]]
        rows, poses = [None] * n, [None] * n
        pending = [None] * C

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
            poses[i] = pose
            rows[i] = evaluate.pack_state(i, pose, np.asarray(pairs[i]["relt_pose"], np.float64), r.num_inliers, r.num_mutual, r.num_inlier_ind,
                                          r.scales_used, data_s, a.elapsed_time(b) / 1e3, [0.0, 0.0, 0.0],
                                          cfg.test.rte_thresh, cfg.test.rre_thresh)
            pending[c] = None

        for i in range(n):
            t0 = time.perf_counter()
            with t.cuda.stream(self.prep_stream):
                src, tgt, aligned, _, _ = self._prepare(tickets[i], voxel_size, replay_rng)
                perm_s, perm_t = [], []
                for _ in range(S):        # models/patch_embedder.py:96, order scale0-src, scale0-tgt, scale1-src, ...
                    perm_s.append(np.random.choice(src.shape[0], src.shape[0], replace=False).astype(np.int32))
                    perm_t.append(np.random.choice(tgt.shape[0], tgt.shape[0], replace=False).astype(np.int32))
                seed = int(np.random.randint(0, 2**31 - 1))
                d_ps = t.from_numpy(np.stack(perm_s)).to(f"cuda:{self.device}", non_blocking=True)
                d_pt = t.from_numpy(np.stack(perm_t)).to(f"cuda:{self.device}", non_blocking=True)
                ready = t.cuda.Event()
                ready.record(self.prep_stream)
            if i + depth < n:
                tickets.append(self.pf.submit(pairs[i + depth]["src_path"], pairs[i + depth]["tgt_path"]))
            data_s = time.perf_counter() - t0
            c = i % C
            harvest(c)
            st = self.streams[c]
            st.wait_event(ready)
            for x in (src, tgt, d_ps, d_pt):
                x.record_stream(st)
            with t.cuda.stream(st):
                a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                a.record(st)
                self.ctxs[c].register_pair_async(src, tgt, aligned, d_ps, d_pt, seed, self.results[c])
                b.record(st)
            pending[c] = (i, a, b, data_s)
        for c in range(C):
            harvest(c)
        return np.stack(rows) if n else np.zeros((0, evaluate.STATE_W)), poses
