#!/usr/bin/env python
"""bench.py -- registered pairs/s (+ p50 ms/pair) of the MI355X-native BUFFER-X hot path.

Workload = BASELINE.json configs[1]: 3DMatch-like pairs, 3 scales, 5000 FPS keypoints, 1024 points/patch,
RANSAC + refinement, on synthetic pairs (N ~ U[20k, 60k] points per cloud, seeded) with seeded random weights
(no datasets / checkpoints exist offline).  A "step" is one pair through bx_register_pair on every rank
(weak scaling: each GPU processes `steps` pairs; pairs are independent, the only collective is one all-gather
of 72-byte result records).  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--inflight C]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel = the Desc conv stack on the f32
matrix cores, HIP-event timed inside the timed region), "roofline_neighbour_gather" (the HBM-bound kernel the
north-star names), "stages_ms", "cpu_baseline" (oracle port on the host cores, bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32
PEAK_HBM_GBS = 8000.0            # HBM3E spec (6.29 TB/s measured achievable)
DESC_CONV_MMAC_PER_PATCH = 3.871 + 5.161 + 10.322 + 20.644 + 10.322 + 5.161 + 2.580 + 1.290


def pmc_traffic():
    """HBM bytes per launch of the two roofline kernels from the committed rocprofv3 PMC passes of THIS command
    (profiles/r01_pmc_traffic.json, derived by tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
    --pmc runs as MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the process; None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)
    except Exception:
        return None


def make_inputs(bx, oracle_perm, n_pairs, base_seed, S, workload="3dmatch"):
    """Distinct seeded synthetic pairs with N ~ U[20k, 60k]; workload "kitti": outdoor LiDAR-like clouds of
    ~75k points aligned to the global z axis (C3; the generator's densest sampling)."""
    pairs = []
    for i in range(n_pairs):
        seed = base_seed + i
        n = int(np.random.default_rng(1000 + seed).integers(20000, 60001))
        p = bx.synth.make_pair(seed, "indoor", n_target=n) if workload == "3dmatch" else bx.synth.make_pair(seed, "outdoor", voxel=0.02)
        rng = np.random.default_rng(seed)
        # permutations: any permutation is a valid stand-in for np.random.choice(N, N, replace=False)
        p["perm_src"] = np.stack([rng.permutation(len(p["src"])).astype(np.int32) for _ in range(S)])
        p["perm_tgt"] = np.stack([rng.permutation(len(p["tgt"])).astype(np.int32) for _ in range(S)])
        p["seed"] = seed
        pairs.append(p)
    return pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--inflight", type=int, default=8, help="pairs in flight per GPU (contexts / HIP streams)")
    ap.add_argument("--lane", type=int, default=0, help="bx_lane mode ordering the pairs in flight: 0 none, 1 whole main phase, 2 conv stacks")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic pairs per rank (cycled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--num-fps", type=int, default=5000)
    ap.add_argument("--ppp", type=int, default=1024)
    ap.add_argument("--scales", type=int, default=3)
    ap.add_argument("--workload", choices=["3dmatch", "kitti"], default="3dmatch",
                    help="3dmatch = BASELINE configs[1] (the headline metric); kitti = configs[2] geometry and match parameters (informational)")
    args = ap.parse_args()
    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues: with more streams than queues two pairs share
    # a queue and run strictly one after the other (measured: 4 pairs in flight were SLOWER than 3).  One queue per pair in flight
    # (+ the null stream and the copy queue); read once when the HIP runtime initialises, i.e. before the first torch.cuda call.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, args.inflight + 4)))

    import torch
    import torch.distributed as dist
    import bufferx_amd as bx
    from bufferx_amd import lib, dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # test hooks (never set by the driver): BX_DIST_BACKEND=gloo + BX_BENCH_SAME_GPU=1 run the N > 1 code path on a 1-GPU box
    backend = os.environ.get("BX_DIST_BACKEND", "nccl")
    if os.environ.get("BX_BENCH_SAME_GPU"):
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    dev = f"cuda:{local}"
    coll_dev = dev if backend == "nccl" else None      # RCCL collectives take device tensors, gloo host tensors

    cfg = bx.make_cfg("3DMatch" if args.workload == "3dmatch" else "KITTI")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = args.num_fps, args.ppp, args.scales
    cfg.patch.search_radius_thresholds = [5, 2, 0.5][:args.scales]
    S, K, P = args.scales, args.num_fps, args.ppp
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))

    pairs = make_inputs(bx, None, args.distinct, 100 + 1000 * rank, S, args.workload)
    dpairs = []
    for p in pairs:
        dpairs.append(dict(src=torch.from_numpy(p["src"]).to(dev), tgt=torch.from_numpy(p["tgt"]).to(dev),
                           perm_src=torch.from_numpy(p["perm_src"]).to(dev), perm_tgt=torch.from_numpy(p["perm_tgt"]).to(dev),
                           seed=p["seed"], aligned=p["aligned_z"], n=(len(p["src"]), len(p["tgt"]))))
    C = max(1, args.inflight)
    ctxs = [lib.Context(cfg, max_points=max(60000, max(max(len(p['src']), len(p['tgt'])) for p in pairs)), device=local, packed_weights=pw) for _ in range(C)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(C)]
    lane = lib.Lane(args.lane) if args.lane and C > 1 else None
    for cx in ctxs:
        cx.attach_lane(lane)
    results = [ctxs[i].new_result() for i in range(C)]

    def run(n_steps, timed):
        """Round-robin the pairs over the contexts; each context's stream serialises its own pairs."""
        lat, recs, evs = [], [], []
        for step in range(n_steps):
            c = step % C
            ctx, st = ctxs[c], streams[c]
            if step >= C:   # the context is busy with pair step-C: wait for it and harvest the result
                st.synchronize()
                a, b, sid, pid = evs[step - C]
                lat.append(a.elapsed_time(b))
                r = results[c]
                recs.append(D.pack_record(sid, np.array(r.pose), r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used, lat[-1]))
            dp = dpairs[step % len(dpairs)]
            with torch.cuda.stream(st):
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record(st)
                ctx.register_pair_async(dp["src"], dp["tgt"], dp["aligned"], dp["perm_src"], dp["perm_tgt"], dp["seed"], results[c])
                b.record(st)
            evs.append((a, b, rank + world * step, step % len(dpairs)))
        for step in range(max(0, n_steps - C), n_steps):
            c = step % C
            streams[c].synchronize()
            a, b, sid, pid = evs[step]
            lat.append(a.elapsed_time(b))
            r = results[c]
            recs.append(D.pack_record(sid, np.array(r.pose), r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used, lat[-1]))
        return lat, recs

    run(C, False)               # every context once (first-use costs: code objects, function attributes), before the W warm-up steps
    torch.cuda.synchronize()
    run(args.warmup, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat, recs = run(args.steps, True)
    allrec = D.gather_records(np.stack(recs), args.steps * world, device=coll_dev)   # the ONE collective (RCCL all-gather)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    assert len(allrec) == args.steps * world
    # Kernel-quality pass: the same workload, ONE pair in flight, hipEvents around every stage on the kernels' own
    # stream (bx_profile_*).  Kept out of the throughput region because with several pairs in flight a kernel's
    # event-to-event time includes the other pairs' kernels it shares the GPU with.
    stages = {}
    NPROF = min(len(dpairs), 4)
    ctxs[0].profile_enable(True)
    with torch.cuda.stream(streams[0]):
        for i in range(NPROF):
            dp = dpairs[i]
            ctxs[0].register_pair_async(dp["src"], dp["tgt"], dp["aligned"], dp["perm_src"], dp["perm_tgt"], dp["seed"], results[0])
            streams[0].synchronize()
    for k, (ms, n) in ctxs[0].profile_read().items():
        stages[k] = [ms, n]
    ctxs[0].profile_enable(False)

    if rank == 0:
        total_pairs = args.steps * world
        value = total_pairs / dt
        # registration quality on the synthetic pairs of rank 0 (sanity: the timed work is real registration)
        ok = 0
        for r in recs:
            u = D.unpack_record(r)
            pid = ((u["pair_id"] - rank) // world) % len(pairs)
            rre, rte = bx.synth.pose_error(u["pose"], pairs[pid]["T_gt"])
            ok += int(rre < cfg.test.rre_thresh and rte < cfg.test.rte_thresh)
        nmean = float(np.mean([dp["n"][0] + dp["n"][1] for dp in dpairs]) / 2)
        # --- dominant kernel: Desc conv stack (8 MFMA launches per cloud per scale)
        conv_ms, conv_n = stages.get("desc_conv", (0.0, 0))
        flops_per_stack = 2.0 * DESC_CONV_MMAC_PER_PATCH * 1e6 * K
        roof = None
        pmc = pmc_traffic()
        if conv_n:
            ach = flops_per_stack / (conv_ms / conv_n * 1e-3) / 1e12
            roof = {"kernel": "conv_kernel<...> x8 (Cylindrical_Net stack, v_mfma_f32_16x16x4_f32)", "bound": "mfma",
                    "achieved": round(ach, 3), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MATRIX_TFLOPS, 4),
                    "traffic": pmc["desc_conv_stack_bytes_per_launch"] if pmc else None,
                    "traffic_note": "HBM bytes per stack launch, rocprofv3 PMC (profiles/r01_pmc_traffic.json); algorithmic in+out maps = 3.27e9",
                    "avg_launch_ms": round(conv_ms / conv_n, 4), "launches": conv_n,
                    "algorithmic_flops_per_launch": flops_per_stack,
                    "note": "hipEvent-timed, one pair in flight, %d pairs right after the timed region" % NPROF}
        # --- the HBM-bound kernel the north-star names: neighbour gather.  Algorithmic bytes per launch:
        #     12N (cloud) + 12K (queries) + 4KP (ball_query idx) + 12KP (grouped xyz); the launch writes both outputs.
        #     "kernel" = ball_query_kernel alone (its own hipEvent bracket); "stage" adds the per-launch grid build.
        ng_ms, ng_n = stages.get("neighbour_gather_query_kernel", (0.0, 0))
        st_ms, st_n = stages.get("neighbour_gather", (0.0, 0))
        roof_ng = None
        if ng_n:
            nbytes = 12.0 * nmean + 12.0 * K + 4.0 * K * P + 12.0 * K * P
            ach = nbytes / (ng_ms / ng_n * 1e-3) / 1e9
            roof_ng = {"kernel": "ball_query_kernel", "bound": "hbm", "achieved": round(ach, 2), "peak": PEAK_HBM_GBS,
                       "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                       "traffic": pmc["ball_query_bytes_per_launch"] if pmc else None,
                       "avg_launch_ms": round(ng_ms / ng_n, 4), "launches": ng_n, "algorithmic_bytes_per_launch": nbytes,
                       "stage_avg_ms_incl_grid_build": round(st_ms / st_n, 4) if st_n else None}
        out = {
            "metric": "registered pairs/sec + p50 ms/pair, 3DMatch 5k-FPS 3-scale, 1/2/4/8 MI355X",
            "value": round(value, 4), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "p50_ms_per_pair": round(float(np.median(lat)), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("3DMatch-like synthetic pairs, %d scales, %d FPS keypoints, %d pts/patch, RANSAC+refine, "
                                    "N~U[20k,60k] pts/cloud (BASELINE configs[1])" % (S, K, P)) if args.workload == "3dmatch" else
                                   ("KITTI-like synthetic outdoor pairs (aligned z, confidence 1.0 = 50k RANSAC iterations, no refinement), "
                                    "%d scales, %d FPS keypoints, %d pts/patch (BASELINE configs[2] geometry; informational)" % (S, K, P)),
                       "pairs_in_flight_per_gpu": C, "parallelism": "pair-sharded x%d, one all-gather of 72 B records" % world,
                       "weights": "seeded random (reference snapshot layout)", "mean_points_per_cloud": nmean},
            "registered_ok": "%d/%d" % (ok, len(recs)),
            "roofline": roof, "roofline_neighbour_gather": roof_ng,
            "stages_ms_per_pair": {k: round(v[0] / NPROF, 3) for k, v in stages.items() if v[1]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(bx, cfg, pw, pairs[0])
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(bx, cfg, pw, pair):
    """Oracle (CPU port, OpenMP over keypoints) on a bounded sample of the SAME workload: the same pair of clouds,
    same patch size and radius schedule, but 160 keypoints and 1 scale; scaled to a full pair by the number of
    (keypoint, scale) patches, which dominates the cost (convolutions + neighbour search are linear in it)."""
    from oracle import pipeline as PL
    import copy
    cores = os.cpu_count()
    c2 = copy.deepcopy(cfg)
    Ks = 160
    c2.patch.num_fps, c2.patch.num_scales = Ks, 1
    c2.patch.search_radius_thresholds = [cfg.patch.search_radius_thresholds[0]]
    c2.patch.num_points_radius_estimate = 256
    t0 = time.perf_counter()
    PL.register_pair(pair["src"], pair["tgt"], pw, c2, pair["aligned_z"], 1)
    t = time.perf_counter() - t0
    scale = (cfg.patch.num_fps * cfg.patch.num_scales) / float(Ks)
    return {"value": round(1.0 / (t * scale), 6), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle (C, -O2, OpenMP) register_pair on one synthetic pair (N=%d/%d), %d keypoints x 1 scale, "
                      "%d pts/patch: %.1f s; extrapolated x%.1f by patch count to %d keypoints x %d scales"
                      % (len(pair["src"]), len(pair["tgt"]), Ks, cfg.patch.num_points_per_patch, t, scale,
                         cfg.patch.num_fps, cfg.patch.num_scales),
            "sample_seconds": round(t, 2)}


if __name__ == "__main__":
    main()
