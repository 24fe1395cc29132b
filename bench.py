#!/usr/bin/env python
"""bench.py -- registered pairs/s (+ p50 ms/pair) of the MI355X-native BUFFER-X hot path.

Headline workload = BASELINE.json configs[1]: 3DMatch-like pairs, 3 scales, 5000 FPS keypoints, 1024 points/patch,
RANSAC + refinement, on synthetic pairs (N ~ U[20k, 60k] points per cloud, seeded) with seeded random weights
(no datasets / checkpoints exist offline).  The pairs are noise-free PARTIAL-overlap fragments cut from one voxelised
sample of a scene (synth.make_pair(shared=True)): with randomly initialised weights these register, so the
data-dependent stages (CostNet on m matches, consensus on M, RANSAC on C) do representative work and `registered_ok`
means something.  A "step" is one pair through bx_register_pair on every rank (weak scaling: each GPU processes
`steps` pairs; pairs are independent, the only collective is one all-gather of 192-byte float64 result records).
Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--inflight C] [--workload 3dmatch|3dlomatch|3dmatch-noisy|kitti|tiers]
  N > 1: either  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
         or      python bench.py --gpus N   (no WORLD_SIZE in the environment: bench.py spawns its own N ranks)

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel = the Desc conv stack on the f32 matrix
cores, HIP-event timed), "roofline_costnet", "roofline_neighbour_gather" (the HBM-bound stage the north-star names;
frac = algorithmic bytes / STAGE time incl. all grid work), "stages_ms_per_pair", "work" (m / M / C / RANSAC
iterations actually seen), "cpu_baseline" (+ "cpu_baseline_neighbour": the reference's own nanoflann radius search).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32
PEAK_HBM_GBS = 8000.0            # HBM3E spec (6.29 TB/s measured achievable)
DESC_CONV_MMAC_PER_PATCH = 3.871 + 5.161 + 10.322 + 20.644 + 10.322 + 5.161 + 2.580 + 1.290
# what the kernels issue on the matrix pipe for the Desc stack, per form (bx_params.desc_conv_form):
#  winograd43 (default, ALL eight layers): 36 planes x 32 tile rows per 3.2 units = 360 / 1260 of the direct MACs (round 5: a workgroup
#      item is 32 consecutive tile rows; round 4 issued 36 x 32 per THREE units = 384 / 1260, two of every 32 rows being padding -- the same
#      kernel time now shows a 6 % LOWER issued-flop fraction although nothing got slower: the algorithmic rate below is the comparable one)
#  winograd22: layers 0..5 (>= 64 output channels) 16 planes x 40 tile rows per two units = 640 / 1260, layers 6, 7 direct
DESC_EXECUTED_MMAC_PER_PATCH = {
    "winograd43": DESC_CONV_MMAC_PER_PATCH * 360.0 / 1260.0,
    # mixed tiles (round 6): 36 planes on the F(4x4) tiles of the map rows 0..3 + 30 on the F(3x4) tiles of the rows 4..6 = 66 plane-rows per
    # column block, 330 per unit (the 24 MFMAs a wave skips are really not issued)
    "winograd43m": DESC_CONV_MMAC_PER_PATCH * 330.0 / 1260.0,
    "winograd22": (3.871 + 5.161 + 10.322 + 20.644 + 10.322 + 5.161) * 640.0 / 1260.0 + 2.580 + 1.290,
    "direct": DESC_CONV_MMAC_PER_PATCH,
}
COSTNET_MMAC_PER_MATCH = 80.0    # the reference's CostNet on the materialised 20-shift cost volume
# what the shipped kernels execute: layer 0 collapsed to its P - Q form (k_cost.hip: 1.42 MMAC of binary64 VALU work instead of the
# 26.87 MMAC fp32 convolution of the volume); the remaining 53.1 MMAC of layers 1..9 are f32 MFMA work
COSTNET_L0_MMAC, COSTNET_L0_COLLAPSED_MMAC = 26.873, 1.42

WORKLOADS = {
    "3dmatch": ("3DMatch", "3DMatch-like synthetic pairs (noise-free partial-overlap fragments of one voxelised scene sample), "
                           "%d scales, %d FPS keypoints, %d pts/patch, RANSAC+refine, N~U[20k,60k] pts/cloud (BASELINE configs[1])"),
    "3dmatch-noisy": ("3DMatch", "3DMatch-like synthetic pairs (independently sampled, jittered fragments: random weights cannot "
                                 "register these), %d scales, %d FPS keypoints, %d pts/patch, RANSAC+refine, N~U[20k,60k] (round-1 workload)"),
    "3dlomatch": ("3DLoMatch", "3DLoMatch-like synthetic pairs (as 3dmatch, overlap 10-30 %%), %d scales, %d FPS keypoints, %d pts/patch, "
                               "RANSAC+refine, N~U[20k,60k] pts/cloud (the low-overlap half of BASELINE configs[3])"),
    "kitti": ("KITTI", "KITTI-like synthetic outdoor pairs (two LiDAR sweeps, aligned z, confidence 1.0 = 50k RANSAC iterations, no "
                       "refinement), %d scales, %d FPS keypoints, %d pts/patch (BASELINE configs[2] geometry)"),
    "tiers": ("TIERS_hetero", "TIERS_hetero-like pairs (dense 128-ring sweep of ~100k points vs its 64-ring subset of ~50k points re-posed, "
                              "range <= 20 m, outdoor match parameters, early exit ON with 50 inliers), %d scales, %d FPS keypoints, %d pts/patch (BASELINE configs[4] geometry)"),
}


def profile_json(name, files=()):
    """Numbers that cannot be read from inside the process (rocprofv3 PMC passes of THIS command, committed under profiles/).
    A profile file carries the sha256 of every HIP source it was measured on ("hip_sources_sha", tools/src_sha.py); when one of the
    `files` the quoted kernel lives in has changed since, the number no longer belongs to the shipped kernel: "_stale" is set and
    the caller reports null instead of a stale constant."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from src_sha import sha_map
        now = sha_map()
    except Exception:
        now = {}
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, name))) as f:
                d = json.load(f)
        except Exception:
            continue
        d["_file"] = "profiles/%s_%s.json" % (rnd, name)
        then = d.get("hip_sources_sha") or {}
        d["_stale"] = (not then) or any(then.get(k) != now.get(k) for k in files)
        return d
    return None


def make_pair(bx, workload, seed):
    if workload == "3dmatch":
        n = int(np.random.default_rng(1000 + seed).integers(20000, 60001))
        return bx.synth.make_pair(seed, "indoor", n_target=n, shared=True)
    if workload == "3dlomatch":
        rng = np.random.default_rng(2000 + seed)
        n = int(rng.integers(20000, 60001))
        return bx.synth.make_pair(seed, "indoor", n_target=n, shared=True, overlap=float(rng.uniform(0.1, 0.3)))
    if workload == "3dmatch-noisy":
        n = int(np.random.default_rng(1000 + seed).integers(20000, 60001))
        return bx.synth.make_pair(seed, "indoor", n_target=n)
    if workload == "kitti":
        return bx.synth.make_pair(seed, "outdoor", voxel=0.02)
    if workload == "tiers":
        return bx.synth.make_tiers_pair(seed)
    raise ValueError(workload)


def make_inputs(bx, n_pairs, base_seed, S, workload):
    """Distinct seeded synthetic pairs; the same list on every rank."""
    pairs = []
    for i in range(n_pairs):
        seed = base_seed + i
        p = make_pair(bx, workload, seed)
        rng = np.random.default_rng(seed)
        # permutations: any permutation is a valid stand-in for np.random.choice(N, N, replace=False)
        p["perm_src"] = np.stack([rng.permutation(len(p["src"])).astype(np.int32) for _ in range(S)])
        p["perm_tgt"] = np.stack([rng.permutation(len(p["tgt"])).astype(np.int32) for _ in range(S)])
        p["seed"] = seed
        pairs.append(p)
    return pairs


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N ranks of this script through buffer-x_amd/dist.py::spawn_ranks (fail-fast:
    the first rank that dies ends the job with its stderr tail and a non-zero exit code instead of leaving the others in the
    collective until its timeout; per-rank logs kept), rank 0's JSON line passed through."""
    import bufferx_amd  # noqa: F401  (registers the hyphenated package directory)
    from bufferx_amd import dist as D
    rc, out0, log_dir = D.spawn_ranks(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      log_dir=os.environ.get("BX_RANK_LOG_DIR"))
    sys.stdout.write(out0)
    sys.stdout.flush()
    if rc == 0:
        # rank 0's stderr carries the progress notes of the run: pass them on like a single-process run would
        try:
            sys.stderr.write(open(os.path.join(log_dir, "rank0.err")).read())
        except OSError:
            pass
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--inflight", type=int, default=16, help="pairs in flight per GPU (contexts / HIP streams); measured 8 / 12 / 16 / 24 / 32: 50.8 / 51.9 / 52.6 / 49.7 / 50.6 pairs/s")
    ap.add_argument("--lane", type=int, default=0, help="bx_lane mode ordering the pairs in flight: 0 none, 1 whole main phase, 2 conv stacks")
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic pairs (cycled; the same list on every rank); N of every cloud is drawn from U[20k, 60k]")
    ap.add_argument("--desc-conv", default=None, choices=["winograd43", "winograd22", "direct", "winograd43m"], help="bx_params.desc_conv_form (default: the library default)")
    ap.add_argument("--pose-conv", default=None, choices=["winograd43", "winograd22", "direct"], help="bx_params.pose_conv_form")
    ap.add_argument("--cost-l0", default=None, choices=["collapsed", "direct"], help="bx_params.cost_l0_form")
    ap.add_argument("--inflight-sweep", default="1,2,4,8,16", help="pairs in flight of the throughput-vs-latency sweep after the timed region ('' = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-call", action="store_true", help="early-exit workloads: bx_register_pair (every launch of every scale enqueued, device-side skip) instead of the two-call form with the exit decision on the host")
    ap.add_argument("--e2e-pairs", type=int, default=96, help="pairs of the file-driven end-to-end measurement per RNG mode (0 = skip; N = 1 only); 96 = the step count of the hot-path measurement it is compared with (round 4 ran 24: fill and drain of 16 pairs in flight were a third of that run)")
    ap.add_argument("--num-fps", type=int, default=5000)
    ap.add_argument("--latency-tiles", type=int, default=2, help="keypoint tiles of the latency-form measurement (0/1 = skip it)")
    ap.add_argument("--keypoint-tiles", type=int, default=0, help="bx_params.keypoint_tiles of the MAIN contexts (0 / 1 = throughput form; >= 2 = latency form: FPS in tiles on the context's own stream, same results bit for bit)")
    ap.add_argument("--ppp", type=int, default=1024)
    ap.add_argument("--scales", type=int, default=3)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="3dmatch",
                    help="3dmatch = BASELINE configs[1] (the headline metric); kitti = configs[2]; tiers = configs[4] (early exit)")
    ap.add_argument("--dump-records", default=None, help="write the gathered float64 records [pairs, 24] to this .npy (rank 0)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues: with more streams than queues two pairs share
    # a queue and run strictly one after the other (measured: 4 pairs in flight were SLOWER than 3).  One queue per pair in flight
    # (+ the null stream and the copy queue); read once when the HIP runtime initialises, i.e. before the first torch.cuda call.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, args.inflight * (3 if args.keypoint_tiles > 1 else 1) + 4)))

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
    affinity = None
    if world > 1:
        # one process per GPU: host threads next to the GPU (NUMA node from sysfs, else an even split of the CPUs); N = 1 is untouched
        try:
            pr = torch.cuda.get_device_properties(local)
            pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pci = None
        affinity = D.bind_rank_to_gpu(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), pci)
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("BX_DIST_TIMEOUT_S", "180")))     # a rank that never arrives is an error, not a hang
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    dev = f"cuda:{local}"
    coll_dev = dev if backend == "nccl" else None      # RCCL collectives take device tensors, gloo host tensors

    dataset, wl_text = WORKLOADS[args.workload]
    cfg = bx.make_cfg(dataset)
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = args.num_fps, args.ppp, args.scales
    cfg.patch.search_radius_thresholds = [5, 2, 0.5][:args.scales]
    if args.workload == "tiers":
        cfg.match.enable_early_exit = True      # config/outdoor_config.py:76 (the TIERS configs inherit it)
        cfg.match.early_exit_min_inliers = 50
    S, K, P = args.scales, args.num_fps, args.ppp
    if args.keypoint_tiles > 1:
        cfg.test.keypoint_tiles = args.keypoint_tiles
    # arithmetic forms: explicit configuration (bx_params), echoed by every bx_result and checked in harvest() -- never the environment
    for key, val in (("desc_conv", args.desc_conv), ("pose_conv", args.pose_conv), ("cost_l0", args.cost_l0)):
        if val:
            cfg.arith[key] = val
    forms = bx.config.arith_of(cfg)
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))

    pairs = make_inputs(bx, args.distinct, 100, S, args.workload)
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

    def run(n_steps, depth=C, ctxs=ctxs):
        """Round-robin the pairs over `depth` contexts; each context's stream serialises its own pairs.  Global pair id of step s on
        rank r = r + world * s; it selects the synthetic pair (id mod distinct), so any world size processes the same pair list."""
        lat, recs, evs = [], [], []

        def harvest(step):
            c = step % depth
            streams[c].synchronize()
            th = time.perf_counter()
            a, b, gid = evs[step]
            lat.append(a.elapsed_time(b))
            r = results[c]
            if r.status != 0:       # device-side failure bits (e.g. the bounded spin of the multi-workgroup FPS timed out)
                raise RuntimeError("pair %d: bx_result.status = 0x%x" % (gid, r.status))
            if lib.forms_of_result(r) != forms:
                raise RuntimeError("pair %d ran in %s, configured %s" % (gid, lib.forms_of_result(r), forms))
            pose = np.array(r.pose, np.float64).reshape(4, 4)
            if cfg.test.pose_refine is True:
                pose = pose.astype(np.float32)
            recs.append(D.pack_record(gid, pose, r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used, lat[-1], r.ransac_iters))
            host_s[0] += time.perf_counter() - th

        for step in range(n_steps):
            c = step % depth
            ctx, st = ctxs[c], streams[c]
            if step >= depth:   # the context is busy with pair step-depth: wait for it and harvest the result
                harvest(step - depth)
            gid = rank + world * step
            dp = dpairs[gid % len(dpairs)]
            th = time.perf_counter()
            with torch.cuda.stream(st):
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record(st)
                ctx.register_pair_async(dp["src"], dp["tgt"], dp["aligned"], dp["perm_src"], dp["perm_tgt"], dp["seed"], results[c])
                b.record(st)
            host_s[0] += time.perf_counter() - th
            evs.append((a, b, gid))
        for step in range(max(0, n_steps - depth), n_steps):
            harvest(step)
        return lat, recs

    def run_two_calls(n_steps, depth=C, ctxs=ctxs):
        """Early-exit configurations: every pair as bx_register_pair_begin -> (host reads the exit decision) -> bx_register_pair_finish,
        the way the reference takes the decision (models/BUFFERX.py:424-457), so that a pair that leaves at scale 0 never enqueues the
        launches of the later scales.  Pairs differ in length now, so the contexts are served as they come free (an event poll), not
        round-robin.  Same records as run(): global pair id of step s on rank r = r + world * s."""
        lat, recs = [], []
        free = list(range(depth))[::-1]
        busy = {}            # context -> [phase, event a, event of the phase's end, gid]
        nxt = done = 0
        while done < n_steps:
            moved = False
            for c in list(busy):
                ph, a, e, gid = busy[c]
                if not e.query():
                    continue
                moved = True
                th = time.perf_counter()
                if ph == 1:
                    with torch.cuda.stream(streams[c]):
                        ctxs[c].register_pair_finish_async(int(flags[c][0]), results[c])
                        b = torch.cuda.Event(enable_timing=True)
                        b.record(streams[c])
                    busy[c] = [2, a, b, gid]
                else:
                    lat.append(a.elapsed_time(e))
                    r = results[c]
                    if r.status != 0:
                        raise RuntimeError("pair %d: bx_result.status = 0x%x" % (gid, r.status))
                    if lib.forms_of_result(r) != forms:
                        raise RuntimeError("pair %d ran in %s, configured %s" % (gid, lib.forms_of_result(r), forms))
                    pose = np.array(r.pose, np.float64).reshape(4, 4)
                    if cfg.test.pose_refine is True:
                        pose = pose.astype(np.float32)
                    recs.append(D.pack_record(gid, pose, r.num_inliers, r.num_mutual, r.num_inlier_ind, r.scales_used, lat[-1], r.ransac_iters))
                    del busy[c]
                    free.append(c)
                    done += 1
                host_s[0] += time.perf_counter() - th
            while free and nxt < n_steps:
                moved = True
                c = free.pop()
                gid = rank + world * nxt
                dp = dpairs[gid % len(dpairs)]
                th = time.perf_counter()
                with torch.cuda.stream(streams[c]):
                    a = torch.cuda.Event(enable_timing=True)
                    e = torch.cuda.Event(enable_timing=True)
                    a.record(streams[c])
                    ctxs[c].register_pair_begin_async(dp["src"], dp["tgt"], dp["aligned"], dp["perm_src"], dp["perm_tgt"], dp["seed"], flags[c])
                    e.record(streams[c])
                busy[c] = [1, a, e, gid]
                nxt += 1
                host_s[0] += time.perf_counter() - th
            if not moved:
                time.sleep(2e-5)
        order = np.argsort([int(r[0]) for r in recs], kind="stable")     # completion order -> pair order (what run() returns)
        return [lat[i] for i in order], [recs[i] for i in order]

    two_calls = bool(cfg.match.get("enable_early_exit", False)) and S > 1 and not args.one_call
    flags = [cx.new_exit_flag() for cx in ctxs]
    run_1 = run
    if two_calls:
        run = run_two_calls
    host_s = [0.0]       # host seconds spent enqueueing (the ~170 launches of a pair) and harvesting (record packing), waits excluded
    run(C)               # every context once (first-use costs: code objects, function attributes), before the W warm-up steps
    torch.cuda.synchronize()
    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    host_s[0] = 0.0
    t0 = time.perf_counter()
    lat, recs = run(args.steps)
    host_ms_per_pair = host_s[0] / max(1, args.steps) * 1e3
    allrec = D.gather_records(np.stack(recs), args.steps * world, device=coll_dev)   # the ONE collective (RCCL all-gather)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_local = dt
    tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # evidence that the collective really saw `world` ranks (outside the timed region): every rank's own wall time and device index,
    # gathered with the same all-gather the records use (RCCL for backend nccl)
    rank_rows = D.gather_rows(np.array([[rank, dt_local, local, torch.cuda.current_device()]], np.float64), world, device=coll_dev)

    assert len(allrec) == args.steps * world
    # the job the driver asked for is the job that ran: every rank arrived in the collective, every pair id is there exactly once
    assert len(rank_rows) == args.gpus and sorted(int(r[0]) for r in rank_rows) == list(range(args.gpus)), \
        "--gpus %d but the all-gather saw ranks %s" % (args.gpus, [int(r[0]) for r in rank_rows])
    assert np.array_equal(np.sort(allrec[:, 0].astype(np.int64)), np.arange(args.steps * world)), "pair ids missing from the gathered records"
    if rank == 0 and args.dump_records:
        np.save(args.dump_records, allrec)
    # Latency at ONE pair in flight (service time of a pair; p50_ms_per_pair above is queueing latency at `inflight` pairs in flight)
    n1 = min(len(dpairs) * 2, 8)
    lat1, rec1 = run(n1, depth=1)
    # throughput vs latency over the number of pairs in flight (short runs right after the timed region; the headline uses --inflight)
    sweep = []
    for d in [int(x) for x in args.inflight_sweep.split(",") if x.strip()]:
        if d < 1 or d > C:
            continue
        nsw = max(2 * d, min(16, 4 * d))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        lsw, _ = run(nsw, depth=d)
        torch.cuda.synchronize()
        sweep.append({"inflight": d, "pairs_per_s": round(nsw / (time.perf_counter() - ts), 3), "p50_ms": round(float(np.median(lsw)), 3), "pairs": nsw})
    # the same in the LATENCY FORM of the whole-pair call (bx_params.keypoint_tiles): FPS on the context's own stream in tiles,
    # the descriptors of a tile computed while the next one is sampled.  Same results bit for bit (checked here).
    lat1t, tiles_same = None, None
    if args.latency_tiles > 1:
        import copy
        cfg_t = copy.deepcopy(cfg)
        cfg_t.test.keypoint_tiles = args.latency_tiles
        ctx_t = lib.Context(cfg_t, max_points=max(60000, max(max(len(p['src']), len(p['tgt'])) for p in pairs)), device=local, packed_weights=pw)
        run_1(2, depth=1, ctxs=[ctx_t])       # (the latency form has no two-call entry)
        lat1t, rec1t = run_1(n1, depth=1, ctxs=[ctx_t])
        tiles_same = bool(all(np.array_equal(a[:22], b[:22]) for a, b in zip(rec1, rec1t)))
        ctx_t.close()
    # Kernel-quality pass: the same workload, ONE pair in flight, hipEvents around every stage on the kernels' own
    # stream (bx_profile_*).  Kept out of the throughput region because with several pairs in flight a kernel's
    # event-to-event time includes the other pairs' kernels it shares the GPU with.
    stages = {}
    NPROF = min(len(dpairs), 4)
    ctxs[0].profile_enable(True)
    prof_scales = []
    with torch.cuda.stream(streams[0]):
        for i in range(NPROF):
            dp = dpairs[i]
            ctxs[0].register_pair_async(dp["src"], dp["tgt"], dp["aligned"], dp["perm_src"], dp["perm_tgt"], dp["seed"], results[0])
            streams[0].synchronize()
            prof_scales.append(int(results[0].scales_used))
    for k, (ms, n) in ctxs[0].profile_read().items():
        stages[k] = [ms, n]
    ctxs[0].profile_enable(False)

    if rank == 0:
        total_pairs = args.steps * world
        value = total_pairs / dt
        # registration quality + data-dependent work over ALL gathered records (every rank's pairs)
        ok = 0
        us = [D.unpack_record(r) for r in allrec]
        for u in us:
            rre, rte = bx.synth.pose_error(np.asarray(u["pose"], np.float64), pairs[u["pair_id"] % len(pairs)]["T_gt"])
            ok += int(rre < cfg.test.rre_thresh and rte < cfg.test.rte_thresh)
        mean_scales = float(np.mean([u["scales_used"] for u in us]))
        mean_M = float(np.mean([u["num_mutual"] for u in us]))
        mean_m = float(np.mean([u["num_mutual"] / max(1, u["scales_used"]) for u in us]))
        mean_C = float(np.mean([u["num_inlier_ind"] for u in us]))
        work = {"mean_m_per_scale": round(mean_m, 1), "mean_M": round(mean_M, 1), "mean_C": round(mean_C, 1),
                "mean_ransac_iters": round(float(np.mean([u["ransac_iters"] for u in us])), 1),
                "mean_ransac_inliers": round(float(np.mean([u["num_inliers"] for u in us])), 1),
                "mean_scales_used": round(mean_scales, 3),
                "early_exit_taken": "%d/%d" % (sum(u["scales_used"] < S for u in us), len(us))}
        nmean = float(np.mean([dp["n"][0] + dp["n"][1] for dp in dpairs]) / 2)
        # launches that did work: with the early exit taken the kernels of the later scales return at once (device-side skip
        # flag) but are still bracketed by events -- per-launch averages are taken over the scales that ran IN THE PROFILED PAIRS
        # (round 4 scaled by the mean over all timed records: with the exit taken in some pairs and not in the profiled ones the
        # launch count was too small and `frac` too large -- profiles/r04_bench_tiers.json said 0.83)
        ran = float(np.mean(prof_scales)) / S
        CONV_SRC = ("k_conv.hip", "k_wino.hip", "k_wino43.hip", "k_wino43v.hip", "wino43_common.h", "bx_common.h")
        pmc = profile_json("pmc_traffic", CONV_SRC)
        pmc_ball = profile_json("pmc_traffic", ("k_ball.hip", "bx_common.h"))
        busy = profile_json("mfma_busy", CONV_SRC)
        busy_cn = profile_json("mfma_busy", CONV_SRC + ("k_cost.hip",))
        fresh = lambda d: d is not None and not d["_stale"]
        stale_note = lambda d: None if d is None else ("%s (%s)" % (d["_file"], "measured on these kernel sources" if not d["_stale"] else "STALE: the kernel sources changed since that profile, value withheld"))
        # --- dominant kernel: Desc conv stack (8 MFMA launches per cloud per scale)
        conv_ms, conv_n = stages.get("desc_conv", (0.0, 0))
        # units of one stack launch: the throughput form of the whole-pair call runs BOTH clouds of a scale as one stack of 2 K units
        # (round 6, bx_api.hip::desc_stack_pair; BX_DESC_BATCH=0 is the measurement hook for one cloud per stack)
        KU = K * (2 if os.environ.get("BX_DESC_BATCH", "1") != "0" and args.keypoint_tiles <= 1 else 1)
        flops_per_stack = 2.0 * DESC_CONV_MMAC_PER_PATCH * 1e6 * KU
        roof = None
        if conv_n:
            conv_n = max(1, int(round(conv_n * ran)))
            ach = flops_per_stack / (conv_ms / conv_n * 1e-3) / 1e12
            form = forms["desc_conv"]
            ex_mmac = DESC_EXECUTED_MMAC_PER_PATCH[form]
            ex_ach = 2.0 * ex_mmac * 1e6 * KU / (conv_ms / conv_n * 1e-3) / 1e12
            roof = {"kernel": {"winograd43": "wino43_kernel<...> x8 (Winograd F(4x4,3x3), items of 32 tile rows; every layer)",
                               "winograd43m": "wino43m_kernel<...> x8 (mixed Winograd tiles: F(4x4,3x3) rows 0..3 + F(3x4,3x3) rows 4..6, items of 16 column blocks; every layer)",
                               "winograd22": "wino_pair_kernel<...> x6 (Winograd F(2x2,3x3), two units per workgroup) + conv_kernel<...> x2",
                               "direct": "conv_kernel<...> x8"}[form] + " (Cylindrical_Net stack, f32 MFMA)",
                    "bound": "mfma",
                    # the roofline statement: flops ISSUED on the matrix pipe / time / peak (<= 1 by construction)
                    "achieved": round(ex_ach, 3), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ex_ach / PEAK_F32_MATRIX_TFLOPS, 4),
                    "executed_flops_per_launch": 2.0 * ex_mmac * 1e6 * KU, "units_per_launch": KU,
                    # the algorithm-adjusted rate: what the reference's convolution multiplies (59.35 MMAC per patch) over
                    # the same time; exceeds the peak when a Winograd form issues fewer multiplications -- NOT a roofline fraction
                    "algorithmic_flops_per_launch": flops_per_stack, "algorithmic_rate_tflops": round(ach, 3),
                    "algorithmic_rate_x_peak": round(ach / PEAK_F32_MATRIX_TFLOPS, 4),
                    "issued_over_direct_macs": round(ex_mmac / DESC_CONV_MMAC_PER_PATCH, 4),
                    "traffic": pmc["desc_conv_stack_bytes_per_launch"] if fresh(pmc) else None,
                    "traffic_note": "HBM bytes per stack launch, rocprofv3 PMC: %s; algorithmic in+out maps = %.3g" % (stale_note(pmc), 3.27e9 * KU / 5000.0),
                    "mfma_busy": busy.get("desc_conv_stack") if fresh(busy) else None,
                    "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), time-weighted over the 8 layers: %s" % stale_note(busy),
                    "avg_launch_ms": round(conv_ms / conv_n, 4), "launches": conv_n,
                    "profiled_pairs_scales_used": prof_scales,
                    "note": "hipEvent-timed on the kernels' stream, one pair in flight, %d pairs right after the timed region; launches = "
                            "event brackets x (scales that ran in those pairs / S)" % NPROF}
        # --- CostNet (cost_l1_kernel + 9 conv_kernel launches + soft-argmax per scale); m = matches of the profiled pairs
        pose_ms, pose_n = stages.get("pose_net", (0.0, 0))
        roof_cn = None
        if pose_n:
            pose_n = max(1, int(round(pose_n * ran)))
            direct0 = forms["cost_l0"] == "direct"
            wino_p = forms["pose_conv"] != "direct"
            # algorithmic work of layers 1..9 as the reference's convolutions run them (53.12 MMAC per match) + layer 0: 26.87 MMAC on the
            # materialised cost volume, or the 1.42 MMAC that remain of it after the algebraic collapse (k_cost.hip)
            alg = COSTNET_MMAC_PER_MATCH if direct0 else COSTNET_MMAC_PER_MATCH - COSTNET_L0_MMAC + COSTNET_L0_COLLAPSED_MMAC
            # issued on the f32 matrix pipe: layers 1..5 as Winograd F(2x2,3x3) (16 planes x padded tile rows), layers 6..9 direct;
            # the collapsed layer 0 is binary64 VALU work
            # (valid F(4x4,3x3): 36 planes x 32 tile rows per G = 2 / 2 / 3 / 3 / 8 units and layer = 16.515 MMAC per match; F(2x2,3x3): 27.263)
            ex = {"winograd43": 16.515 + 1.661, "winograd22": 27.263 + 1.661, "direct": 53.124}[forms["pose_conv"]] + (COSTNET_L0_MMAC if direct0 else 0.0)
            fl = 2.0 * alg * 1e6 * mean_m
            ach = fl / (pose_ms / pose_n * 1e-3) / 1e12
            ex_ach = 2.0 * ex * 1e6 * mean_m / (pose_ms / pose_n * 1e-3) / 1e12
            roof_cn = {"kernel": ("cost_l1_kernel" if direct0 else "cost_l0_kernel (collapsed layer 0, binary64 VALU)") +
                                 ({"winograd43": " + wino43v_kernel x5 (valid Winograd F(4x4,3x3)) + conv_kernel x4", "winograd22": " + wino_pose_kernel x5 (Winograd F(2x2,3x3)) + conv_kernel x4",
                                   "direct": " + conv_kernel x9"}[forms["pose_conv"]]) + " + soft_argmax (CostNet)",
                       "bound": "mfma", "achieved": round(ex_ach, 3),
                       "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": round(ex_ach / PEAK_F32_MATRIX_TFLOPS, 4),
                       "frac_note": "flops ISSUED on the f32 matrix pipe (%.2f MMAC per match) over the time of the whole CostNet / peak" % ex,
                       "algorithmic_rate_tflops": round(ach, 3), "algorithmic_rate_x_peak": round(ach / PEAK_F32_MATRIX_TFLOPS, 4),
                       "flops_note": "algorithmic_* = %.2f MMAC per match (layers 1..9 as direct convolutions + layer 0 %s) over the same time; the "
                                     "reference's own arithmetic (80.0 MMAC per match on the materialised cost volume) in that time = %.1f TFLOP/s"
                                     % (alg, "on the materialised volume" if direct0 else "after its algebraic collapse: 1.42 MMAC",
                                        2.0 * COSTNET_MMAC_PER_MATCH * 1e6 * mean_m / (pose_ms / pose_n * 1e-3) / 1e12),
                       "traffic": None, "mfma_busy": busy_cn.get("costnet") if fresh(busy_cn) else None,
                       "avg_launch_ms": round(pose_ms / pose_n, 4), "launches": pose_n,
                       "algorithmic_flops_per_launch": fl, "mean_matches_per_launch": round(mean_m, 1)}
        # --- the HBM-bound stage the north-star names: neighbour gather.  Algorithmic bytes per (cloud, scale) call:
        #     12N (cloud) + 12K (queries) + 4KP (ball_query idx) + 12KP (grouped xyz).  HEADLINE fraction = bytes / STAGE time (every
        #     launch the stage needs, grid work included, amortised over the calls of a pair); the query kernel alone is also listed.
        ng_ms, ng_n = stages.get("neighbour_gather_query_kernel", (0.0, 0))
        gb_ms, gb_n = stages.get("neighbour_grid_build", (0.0, 0))
        roof_ng = None
        if ng_n:
            nbytes = 12.0 * nmean + 12.0 * K + 4.0 * K * P + 12.0 * K * P
            # every launch of the stage: the batched grid + row-table build (six launches per PAIR, all 2 x S sets at once) and
            # one ball_query_kernel per (cloud, scale); hipEvent brackets on the kernels' stream
            ng_n = max(1, int(round(ng_n * ran)))
            stage_ms = (ng_ms + gb_ms) / ng_n
            ach = nbytes / (stage_ms * 1e-3) / 1e9
            kach = nbytes / (ng_ms / ng_n * 1e-3) / 1e9
            roof_ng = {"kernel": "neighbour-gather STAGE (batched grid + row-table build, ball_query_kernel)", "bound": "hbm",
                       "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                       "traffic": pmc_ball.get("ball_query_bytes_per_launch") if fresh(pmc_ball) else None,
                       "avg_launch_ms": round(stage_ms, 4), "launches": ng_n, "algorithmic_bytes_per_launch": nbytes,
                       "grid_build_ms_per_pair": round(gb_ms / max(gb_n, 1), 4),
                       "query_kernel_avg_ms": round(ng_ms / ng_n, 4), "query_kernel_frac": round(kach / PEAK_HBM_GBS, 4),
                       # beside the SURVEY 8(d) fraction: the same on the bytes the query kernel really MOVES (PMC; the counted hand-over
                       # writes only the real slots of a patch, and no index list)
                       "query_kernel_frac_on_moved_bytes": (round(pmc_ball["ball_query_bytes_per_launch"] / (ng_ms / ng_n * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                                                            if fresh(pmc_ball) and pmc_ball.get("ball_query_bytes_per_launch") else None),
                       "note": "stage time per (cloud, scale) call = (grid build of the pair + its query kernels) / calls; the index list "
                               "(4KP of the algorithmic bytes) is not written by the whole-pair path: nothing reads it"}
        out = {
            "metric": "registered pairs/sec + p50 ms/pair, 3DMatch 5k-FPS 3-scale, 1/2/4/8 MI355X",
            "value": round(value, 4), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            # p50 ms/pair of the metric = SERVICE time of a pair (one in flight, throughput form) -- what the reference's harness times
            # (test.py:145,327-330 runs one pair at a time).  `value` is measured at `inflight` pairs in flight: the two are different
            # operating points and say so; the latency inside the timed region has its own key, as has the single-pair figure.
            "p50_ms_per_pair": round(float(np.median(lat1)), 3),
            "p50_ms_per_pair_inflight1": round(float(np.median(lat1)), 3),
            "p50_definition": {"p50_ms_per_pair": "service time, ONE pair in flight (= p50_ms_per_pair_inflight1); since round 4 -- rounds 1..3 "
                               "printed the in-region latency under this key, now p50_ms_per_pair_queueing_at_inflight",
                               "value": "throughput at %d pairs in flight" % C, "redefined_in_round": 4},
            "p50_ms_per_pair_queueing_at_inflight": {"inflight": C, "p50_ms": round(float(np.median(lat)), 3)},
            "inflight_sweep": sweep,
            "p50_ms_per_pair_latency_form": None if lat1t is None else {
                "p50_ms": round(float(np.median(lat1t)), 3), "keypoint_tiles": args.latency_tiles, "pairs_in_flight": 1,
                "results_identical_to_throughput_form": tiles_same,
                "note": "bx_params.keypoint_tiles: furthest point sampling on the context's own stream in tiles, descriptor work of a tile "
                        "beside the sampling of the next"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_text % (S, K, P),
                       # the metric's second half where the driver keeps it (it stores `config`, not the extra top-level keys):
                       # service time of ONE pair alone, throughput form / latency form (keypoint_tiles)
                       "p50_ms_per_pair": round(float(np.median(lat1)), 3),
                       "p50_ms_per_pair_latency_form": None if lat1t is None else round(float(np.median(lat1t)), 3),
                       "pairs_in_flight_per_gpu": C, "pair_call": ("bx_register_pair_begin + _finish: the early-exit decision on the host, contexts served as they come free" if two_calls else "bx_register_pair"), "arithmetic_forms": forms, "distinct_pairs": len(pairs), "parallelism": "pair-sharded x%d, one all-gather of %d B float64 records" % (world, 8 * D.RECORD),
                       "weights": "seeded random (reference snapshot layout)", "mean_points_per_cloud": nmean,
                       "workload_generator": "synth.make_pair v2 (round 2+: shared=True noise-free partial-overlap fragments; round 1 used "
                                             "independently sampled jittered fragments = --workload 3dmatch-noisy; rates of the two are not comparable)"},
            "collective": dict(collective_evidence(allrec, rank_rows, world, backend if world > 1 else "none (world 1)", args.steps),
                               **({"rank0_cpu_affinity": affinity} if affinity else {})),
            "host_ms_per_pair": round(host_ms_per_pair, 3),
            "host_note": "CPU time of one rank per pair inside the timed region: enqueueing the launches of bx_register_pair + packing the "
                         "result record (stream waits excluded); a rank is host-bound only when this approaches ms_per_step",
            "registered_ok": "%d/%d" % (ok, len(us)), "registered_pairs_per_s": round(value * ok / max(1, len(us)), 4),
            "work": work,
            "roofline": roof, "roofline_costnet": roof_cn, "roofline_neighbour_gather": roof_ng,
            "stages_ms_per_pair": {k: round(v[0] / NPROF, 3) for k, v in stages.items() if v[1]},
        }
        # FPS is bound by its K strictly dependent iterations, not by HBM: report the latency per iteration and the
        # point-update rate (both clouds of a pair run in one launch)
        fps_ms, fps_n = stages.get("fps", (0.0, 0))
        if fps_n:
            it = max(K, cfg.patch.num_points_radius_estimate)
            out["fps"] = {"bound": "dependent-iteration latency", "ms_per_pair": round(fps_ms / fps_n, 3), "iterations": it,
                          "us_per_iteration": round(fps_ms / fps_n / it * 1e3, 3),
                          "point_updates_per_s": round(it * 2.0 * nmean / (fps_ms / fps_n * 1e-3), 0)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_neighbour"] = cpu_baseline(bx, pw, pairs[0], cfg, stages, NPROF)
    for c in ctxs:
        c.close()
    ctxs = []
    if rank == 0:
        if world == 1 and args.e2e_pairs > 0 and args.workload == "3dmatch":
            try:
                out["e2e_pairs_per_s"] = e2e_rate(bx, pw, cfg, args, local, value)
            except Exception as e:      # the end-to-end leg must never take the benchmark line down
                out["e2e_pairs_per_s"] = {"error": repr(e)}
        sp = split_precision_note()
        if sp:
            out["experiments"] = {"split_precision": sp}
        ref = reference_forward_note()
        if ref:
            out["cpu_baseline_reference"] = ref
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def collective_evidence(allrec, rank_rows, world, backend, steps):
    """What the ONE all-gather delivered, read back from the gathered data itself (not from the environment): how many ranks
    contributed records (pair id mod world), how many records each sent, every rank's own wall time of the timed region (gathered
    through the same collective) -> per-rank pairs/s.  A run whose collective silently saw fewer ranks cannot produce this object
    with ranks_contributing == n_gpus."""
    ids = allrec[:, 0].astype(np.int64)
    per_rank = np.bincount(ids % world, minlength=world)
    rates = [steps / float(r[1]) for r in rank_rows]
    return {"backend": backend, "world_size_seen": int(len(rank_rows)), "ranks_contributing": int((per_rank > 0).sum()),
            "records": int(len(allrec)), "records_per_rank": [int(x) for x in per_rank],
            "pair_ids_complete": bool(np.array_equal(np.sort(ids), np.arange(steps * world))),
            "devices_seen": sorted({int(r[3]) for r in rank_rows}),
            "rank_pairs_per_s_min": round(min(rates), 4), "rank_pairs_per_s_max": round(max(rates), 4),
            "rank_wall_s": [round(float(r[1]), 4) for r in rank_rows]}


def split_precision_note():
    """Pointer to the round-3 MEASUREMENT of a split-precision convolution (profiles/r03_split_precision.json); its kernel is not part
    of the library any more (tools/experimental/k_split.hip) and never was part of `value`, `dtype` or the roofline objects."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_split_precision.json")) as f:
            d = json.load(f)
        e = d["layer_error_vs_binary64"]
        return {"file": "profiles/r03_split_precision.json", "shipped": False, "in_library": False,
                "what": "Cylindrical_Net layer 3 (128 -> 128) with 3 x bf16 pieces per fp32 operand, 6 partial products, fp32 accumulation "
                        "(v_mfma_f32_16x16x32_bf16), measured in round 3 against the exact f32 kernels",
                "max_abs_error_vs_binary64": {k: e[k]["max_abs"] for k in ("direct_f32", "winograd_f32", "split_bf16x3")},
                "kernel_us_per_launch": d["kernel_time_us_per_launch_K5000"]}
    except Exception:
        return None


def reference_forward_note():
    """The reference's OWN BufferX.forward (models/BUFFERX.py:257-467, unmodified, un-vendored CUDA ops as numpy stand-ins) timed on
    BASELINE configs[0] in the build container by tools/time_reference.py (it needs /root/reference, which the GPU box does not have):
    read from profiles/, never measured here."""
    for rnd in ("r04",):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_cpu_reference.json" % rnd)) as f:
                d = json.load(f)
            d["file"] = "profiles/%s_cpu_reference.json" % rnd
            return d
        except Exception:
            continue
    return None


def e2e_rate(bx, pw, cfg, args, device, hot_value):
    """Files -> poses: the reference's test loop (test.py:120-200) through buffer-x_amd/harness.py::Runner -- raw scans on disk (binary
    PLY, what the 3DMatch fragments are; four noisy samples per surface point so that the first down-sampling has work) -> native
    prefetch thread -> H2D -> GPU voxel-size analysis + voxel down-sampling + shuffle -> per-scale permutations -> bx_register_pair
    with `inflight` pairs in flight -> evaluation rows (RTE / RRE / success).  Two RNG modes: "reference" replays the loaders' legacy
    NumPy draws call by call in a preparation thread; "device" draws every permutation on the GPU from one host draw per pair."""
    import tempfile
    import shutil
    import torch
    from bufferx_amd import harness
    n_pairs, distinct = int(args.e2e_pairs), 8
    d = tempfile.mkdtemp(prefix="bx_e2e_")
    try:
        rng = np.random.default_rng(0)
        files, raw_n = [], []
        for i in range(distinct):
            p = make_pair(bx, "3dmatch", 500 + i)
            fs = []
            for k, c in (("s", p["src"]), ("t", p["tgt"])):
                raw = np.concatenate([c + rng.normal(0, 0.003, c.shape) for _ in range(4)]).astype("<f4")
                f = os.path.join(d, "%s%d.ply" % (k, i))
                with open(f, "wb") as fh:
                    fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                              "property float z\nend_header\n" % len(raw)).encode("ascii"))
                    fh.write(np.ascontiguousarray(raw).tobytes())
                fs.append(f); raw_n.append(len(raw))
            files.append(dict(src_path=fs[0], tgt_path=fs[1], relt_pose=p["T_gt"]))
        pairs = [files[i % distinct] for i in range(n_pairs)]
        res = {"pairs": n_pairs, "inflight": args.inflight, "mean_raw_points_per_cloud": int(np.mean(raw_n)),
               "note": "files (binary PLY) -> prefetch -> GPU pre-processing -> registration -> evaluation rows, harness.Runner; "
                       "hot-path value of this run = %.2f pairs/s" % hot_value}
        for mode in ("device", "reference"):
            run = harness.Runner(cfg, pw, device=device, inflight=max(1, args.inflight), max_raw_points=max(raw_n), max_points=90000, rng=mode)
            try:
                np.random.seed(0)
                run.run(pairs[:max(4, min(n_pairs, args.inflight))])         # warm-up: every context once
                torch.cuda.synchronize()
                for k in run.timers:
                    run.timers[k] = 0.0
                t0 = time.perf_counter()
                rows, _ = run.run(pairs)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                run.close()
            res[mode] = round(n_pairs / dt, 3)
            res[mode + "_detail"] = {"registered_ok": "%d/%d" % (int(rows[:, 1].sum()), n_pairs),
                                     "prepare_thread_ms_per_pair": round(float(np.mean(rows[:, 8])) * 1e3, 2),
                                     "prepare_thread_breakdown_ms": {k: round(run.timers[k] / n_pairs * 1e3, 2) for k in
                                                                     ("wait_prefetch", "voxel_analysis", "down_sample", "shuffle", "second_sampling_rng", "perm_rng", "perm_upload")},
                                     "registration_thread_ms_per_pair": {k: round(run.timers[k] / n_pairs * 1e3, 2) for k in ("wait_prepared", "harvest_wait", "enqueue")}}
        res["device_over_hot_path"] = round(res["device"] / hot_value, 3)
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(bx, pw, pair, cfg_bench, stages, nprof):
    """(1) The oracle (CPU port, C + OpenMP over keypoints / matches / hypotheses, threads = host cores) on ONE WHOLE pair of
    BASELINE configs[0] -- 1 scale, 512 FPS keypoints, 512 points per patch, RANSAC + refinement -- cut from the same fragments the
    GPU benchmark registers; un-extrapolated (value = configs[0] pairs/s).  `est_workload_pairs_per_s` scales it to the benchmark's
    configuration by (keypoints x scales) patches, for orientation only.
    (2) The reference's OWN neighbour search (cpp_wrappers float nanoflann path, compiled from the reference tree into
    oracle/_ref): batch_nanoflann_neighbors' call sequence for the 2 clouds x 3 radii of the same pair, single thread as in the
    reference (neighbors.cpp:211-332 is a serial loop), beside the GPU stage time of the same step."""
    from oracle import pipeline as PL, oracle as O
    import copy
    cores = os.cpu_count()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    c0 = copy.deepcopy(bx.make_cfg("3DMatch"))
    c0.patch.num_fps, c0.patch.num_points_per_patch, c0.patch.num_scales = 512, 512, 1
    c0.patch.search_radius_thresholds = [5]
    t0 = time.perf_counter()
    cap = {}
    ref = PL.register_pair(pair["src"], pair["tgt"], pw, c0, pair["aligned_z"], 1, cap=cap)
    t = time.perf_counter() - t0
    scale = (cfg_bench.patch.num_fps * cfg_bench.patch.num_scales) / 512.0
    base = {"value": round(1.0 / t, 6), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle (C, -O2, OpenMP, %d threads) register_pair on ONE whole BASELINE configs[0] pair (1 scale, 512 keypoints, "
                      "512 pts/patch, RANSAC+refine; N=%d/%d; %d mutual matches): %.1f s, not extrapolated"
                      % (cores, len(pair["src"]), len(pair["tgt"]), ref[2], t),
            "sample_seconds": round(t, 2), "est_workload_pairs_per_s": round(1.0 / (t * scale), 6),
            "est_note": "x%.1f by (keypoints x scales) to the benchmark configuration; orientation only" % scale}
    # (2) reference nanoflann on the neighbour step of the benchmark configuration
    nb = None
    try:
        K = cfg_bench.patch.num_fps
        kp_s = pair["src"][O.fps(pair["src"], K)]
        kp_t = pair["tgt"][O.fps(pair["tgt"], K)]
        big, bk = (pair["src"], kp_s) if len(pair["src"]) > len(pair["tgt"]) else (pair["tgt"], kp_t)
        nk = min(cfg_bench.patch.num_points_radius_estimate, K)
        radii = [O.radius(big, len(big), bk[:nk], thr) for thr in cfg_bench.patch.search_radius_thresholds]
        t0 = time.perf_counter()
        mc = []
        for r in radii:
            for q, s in ((kp_s, pair["src"]), (kp_t, pair["tgt"])):
                res = O.ref_batch_neighbors(q, s, np.float32(r))
                if res is None:
                    raise RuntimeError("oracle/_ref/libref_neighbors.so absent")
                mc.append(res[0])
        tn = time.perf_counter() - t0
        st_ms = (stages.get("neighbour_gather", (0.0, 0))[0] + stages.get("neighbour_grid_build", (0.0, 0))[0]) / max(1, nprof)
        nb = {"value": round(tn, 3), "unit": "s per pair (neighbour step: 2 clouds x %d radii, %d queries each)" % (len(radii), K),
              "cores": 1, "kind": "reference",
              "sample": "oracle/_ref: cpp_wrappers/cpp_utils nanoflann KD-tree (float, leaf 10) built per cloud + sorted radiusSearch of "
                        "every keypoint + dense padded index matrix = batch_nanoflann_neighbors (neighbors.cpp:211-332), radii %s, "
                        "max neighbours %s" % ([round(float(r), 2) for r in radii], mc),
              "gpu_stage_ms_per_pair": round(st_ms, 3),
              "note": "the reference op returns ALL neighbours sorted by distance; the hot path's ball_query returns the first 1024 in "
                      "index order (different output convention, same radius sets -- tests/test_host_logic.py)"}
    except Exception as e:   # the baseline must never take the benchmark line down
        nb = {"error": str(e)}
    return base, nb


if __name__ == "__main__":
    main()
