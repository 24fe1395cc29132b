#!/usr/bin/env python
"""tools/fps_fuzz.py -- randomised parity run of the furthest-point-sampling kernel (k_fps.hip, the round-6 form that resolves several
samples per cross-workgroup exchange) against the CPU oracle, through the C-ABI (bx_fps).  Not a test (the -m gpu suite holds the
fixed cases): a few minutes of random cloud shapes / sizes / sample counts, one JSON line per case + a summary line.  Shapes are
drawn to stress the resolution: clustered maxima, exact duplicates, distance ties, degenerate Morton buckets, points inside the
origin-skip radius, clouds from one point to a quarter of a million (1 .. 16 workgroups per cloud, every points-per-thread form).

    python tools/fps_fuzz.py --cases 150 --seed 1 > gpurun_out/fps_fuzz.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def draw_cloud(rng, n):
    kind = rng.choice(["uniform", "blobs", "lattice", "duplicates", "plane", "line", "shell", "mixed_scale"])
    f32 = np.float32
    if kind == "uniform":
        xyz = rng.random((n, 3), f32) * f32(rng.uniform(0.5, 40)) - f32(rng.uniform(0, 5))
    elif kind == "blobs":
        nb = int(rng.integers(2, 20))
        c = rng.random((nb, 3), f32) * 8
        xyz = c[rng.integers(0, nb, n)] + rng.standard_normal((n, 3)).astype(f32) * f32(rng.uniform(0.001, 0.2))
    elif kind == "lattice":
        s = max(2, int(round(n ** (1 / 3))) + 1)
        g = np.stack(np.meshgrid(np.arange(s), np.arange(s), np.arange(s), indexing="ij"), -1).reshape(-1, 3).astype(f32)
        xyz = g[rng.permutation(len(g))[:n]] * f32(2.0 ** rng.integers(-4, 2))
        if len(xyz) < n:
            xyz = np.concatenate([xyz, xyz[rng.integers(0, len(xyz), n - len(xyz))]])
    elif kind == "duplicates":
        base = rng.random((max(1, n // int(rng.integers(2, 9))), 3), f32) * 3
        xyz = base[rng.integers(0, len(base), n)]
    elif kind == "plane":
        xyz = rng.random((n, 3), f32) * 10
        xyz[:, 2] = f32(1.5)
    elif kind == "line":
        t = rng.random(n).astype(f32)
        xyz = np.stack([t * 8, t * f32(0.5) + 1, np.full_like(t, 2.0)], 1)
    elif kind == "shell":
        v = rng.standard_normal((n, 3)).astype(f32)
        xyz = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), f32(1e-6)) * f32(rng.uniform(1, 30))
    else:
        xyz = rng.random((n, 3), f32) * 2
        far = rng.random(n) < 0.02
        xyz[far] *= f32(50)
    if rng.random() < 0.3 and n > 4:                       # a few points inside the origin-skip radius (never candidates upstream)
        xyz[rng.integers(0, n, max(1, n // 500))] = (rng.random(3).astype(f32) - f32(0.5)) * f32(1e-3)
    return str(kind), np.ascontiguousarray(xyz[rng.permutation(n)], f32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-n", type=int, default=250000)
    ap.add_argument("--min-n", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=900.0, help="stop drawing cases after this much wall time")
    args = ap.parse_args()
    import bufferx_amd as bx
    from bufferx_amd import lib
    from oracle import oracle as O
    O.lib()
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 5000, 64, 1
    cfg.patch.search_radius_thresholds = [5]
    pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
    ctx = lib.Context(cfg, max_points=args.max_n, device=0, packed_weights=pw)
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    bad, done, samples = 0, 0, 0
    for i in range(args.cases):
        if time.time() - t0 > args.budget_s:
            break
        n = int(np.exp(rng.uniform(np.log(args.min_n), np.log(args.max_n))))
        # the oracle is O(n m): keep a case under ~1.5e9 distance evaluations
        m = int(min(5000, max(1, np.exp(rng.uniform(0, np.log(5000)))), max(1, int(1.5e9 // max(n, 1)))))
        if args.min_n > 1 and rng.random() < 0.5:
            m = int(min(5000, max(1, int(1.5e9 // n))))       # the production sample count on large clouds
        if rng.random() < 0.15:
            m = min(5000, n + int(rng.integers(0, 40)))   # more samples than points now and then
        kind, xyz = draw_cloud(rng, n)
        if rng.random() < 0.25:
            os.environ["BX_FPS_K"] = str(int(rng.choice([1, 2, 4, 6, 8])))
        else:
            os.environ.pop("BX_FPS_K", None)
        hooks = {}
        if rng.random() < 0.25:
            hooks["BX_FPS_PPT"] = str(int(rng.choice([4, 8, 16])))
        if rng.random() < 0.15:
            hooks["BX_FPS_COLOCATE"] = "0"
        for h in ("BX_FPS_PPT", "BX_FPS_COLOCATE"):
            os.environ.pop(h, None)
        os.environ.update(hooks)
        idx, kp = ctx.fps(xyz, m)
        idx, kp = idx.cpu().numpy(), kp.cpu().numpy()
        ref = O.fps(xyz, m)
        ok = bool(np.array_equal(idx, ref) and np.array_equal(kp, xyz[ref]))
        first = int(np.argmax(idx != ref)) if not ok else -1
        print(json.dumps({"case": i, "kind": kind, "n": n, "m": m, "k": os.environ.get("BX_FPS_K", "default"), "hooks": hooks, "ok": ok, "first_diff": first}), flush=True)
        bad += 0 if ok else 1
        done += 1
        samples += m
    print(json.dumps({"summary": True, "cases": done, "mismatching": bad, "samples_compared": samples, "seed": args.seed,
                      "seconds": round(time.time() - t0, 1)}), flush=True)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
