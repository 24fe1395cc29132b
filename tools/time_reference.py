#!/usr/bin/env python
"""tools/time_reference.py -- the reference's OWN BufferX.forward as the CPU baseline (BUILD CONTAINER ONLY: needs /root/reference).

Runs models/BUFFERX.py::BufferX.forward (inference branch, :257-467) unmodified on CPU through tests/golden/ref_harness.py -- the
un-vendored CUDA ops (pointnet2_ops, knn_cuda, torch_batch_svd, kornia, open3d) are the numpy stand-ins the fixtures were minted
with -- on ONE pair of BASELINE configs[0] (1 scale, 512 FPS keypoints, 512 points per patch, RANSAC + refinement; the pair of the
`baseline_cfg0` fixture) and writes profiles/r04_cpu_reference.json.  bench.py quotes that file as `cpu_baseline_reference`
(kind "reference"); it cannot be measured on the GPU box, which has no /root/reference.  Timing convention of the reference:
test.py:24,137-146,327-330 (wall clock around model(data_source))."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_harness as rh  # noqa: E402
import make_golden as MG  # noqa: E402
import bufferx_amd  # noqa: E402


def main():
    name = "baseline_cfg0"
    ns = rh.load_reference()
    ds, pair, seed, ov = MG.case_inputs(name)
    cfg = ns.CFG.make_cfg(ds, "/tmp")
    cfg.stage = "test"
    MG.apply_overrides(cfg, ov)
    model = ns.BX.BufferX(cfg)
    sd = bufferx_amd.weights.synthetic_state_dict(0)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.eval()
    src, tgt = pair["src"], pair["tgt"]
    times = []
    for rep in range(2):
        del rh.PERM_QUEUE[:]
        for i in range(cfg.patch.num_scales):
            rh.PERM_QUEUE.append(rh.make_perm(len(src), seed, 2 * i))
            rh.PERM_QUEUE.append(rh.make_perm(len(tgt), seed, 2 * i + 1))
        rh.RANSAC_STATE.update(seed=seed, calls=0, log=[])
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model({"src_fds_pcd": torch.from_numpy(src), "tgt_fds_pcd": torch.from_numpy(tgt), "is_aligned_to_global_z": pair["aligned_z"]})
        times.append(time.perf_counter() - t0)
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    same = bool(np.allclose(np.asarray(out[0], np.float64), g["pose"], atol=1e-6) and int(out[3]) == int(g["num_mutual"]))
    t = min(times)
    rec = {"value": round(1.0 / t, 5), "unit": "pairs/s", "kind": "reference", "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
           "seconds_per_pair": round(t, 2), "runs_seconds": [round(x, 2) for x in times],
           "sample": "the reference's own BufferX.forward (models/BUFFERX.py:257-467, unmodified) on CPU, ONE pair of BASELINE configs[0] "
                     "(1 scale, 512 FPS keypoints, 512 pts/patch, RANSAC + refinement; N = %d / %d points; the `baseline_cfg0` fixture pair), "
                     "un-vendored CUDA ops as the numpy stand-ins of tests/golden/ref_harness.py; best of two runs, build container "
                     "(no GPU), wall clock around model(data) as test.py:137-146" % (len(src), len(tgt)),
           "reproduces_fixture": same,
           "note": "measured by tools/time_reference.py where /root/reference exists; not re-measured by bench.py (the GPU box has no reference tree)"}
    with open(os.path.join(ROOT, "profiles", "r04_cpu_reference.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
