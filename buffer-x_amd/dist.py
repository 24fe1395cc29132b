"""Multi-GPU plan: pairs are independent units (reference test.py:132-146 carries no state between pairs), so
ranks take pairs round-robin and the ONLY collective is one all-gather of a fixed-size record per pair at
the end (RCCL over xGMI when backend == "nccl"; gloo on CPU for the tests).  SURVEY.md §8e.

The record is float64 (192 bytes per pair: 3 404 pairs of 3DMatch + 3DLoMatch = 654 KB, still one latency-bound collective): an
un-refined RANSAC pose is binary64 (every outdoor configuration), so a float32 record would change its last bits when it crosses
ranks and a sharded run would not be bit-identical to a single-GPU run.  buffer-x_amd/evaluate.py's state rows are float64 for the
same reason; both travel through gather_rows()."""
import os

import numpy as np

# float64: pair_id, pose[16] row-major, num_inliers, num_mutual, num_inlier_ind, scales_used, ransac_iters, model_ms, pose dtype (32 / 64)
RECORD = 24


def shard_indices(n_pairs, rank, world):
    """rank r handles pairs {i : i mod W == r}"""
    return list(range(rank, n_pairs, world))


def pack_record(pair_id, pose, num_inliers, num_mutual, num_inlier_ind, scales_used, model_ms, ransac_iters=0):
    r = np.zeros(RECORD, np.float64)
    pose = np.asarray(pose)
    r[23] = 32 if pose.dtype == np.float32 else 64
    r[0] = pair_id
    r[1:17] = pose.astype(np.float64).reshape(-1)
    r[17:22] = [num_inliers, num_mutual, num_inlier_ind, scales_used, ransac_iters]
    r[22] = model_ms
    return r


def unpack_record(r):
    T = np.asarray(r[1:17], np.float64).reshape(4, 4).copy()
    if r[23] == 32:
        T = T.astype(np.float32)
    return dict(pair_id=int(r[0]), pose=T, num_inliers=int(r[17]), num_mutual=int(r[18]), num_inlier_ind=int(r[19]),
                scales_used=int(r[20]), ransac_iters=int(r[21]), model_ms=float(r[22]))


def gather_rows(local, n_rows, device=None):
    """local: [n_local, W] (float32 or float64) rows of this rank, unit id in column 0 -> [n_rows, W] ordered by id on every
    rank.  ONE all_gather of equal-sized, -1-padded blocks (RCCL over xGMI with backend nccl; gloo on CPU)."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local)
    assert local.ndim == 2 and local.dtype in (np.float32, np.float64)
    W = local.shape[1]
    live = dist.is_available() and dist.is_initialized()
    # BX_DIST_FORCE_COLLECTIVE=1 (test hook): a world of one still goes through the collective, so that the RCCL path
    # (init_process_group("nccl") + all_gather_into_tensor on a device float64 tensor) can be exercised on a 1-GPU box
    if not live or (dist.get_world_size() == 1 and not os.environ.get("BX_DIST_FORCE_COLLECTIVE")):
        return local[np.argsort(local[:, 0], kind="stable")]
    world = dist.get_world_size()
    per = (n_rows + world - 1) // world
    buf = np.full((per, W), -1.0, local.dtype)
    buf[:len(local)] = local
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = torch.empty((world * per, W), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    out = out.cpu().numpy()
    out = out[out[:, 0] >= 0]
    return out[np.argsort(out[:, 0], kind="stable")]


def gather_records(local, n_pairs, device=None):
    """local: float64 [n_local, RECORD] of this rank's pairs -> float64 [n_pairs, RECORD] ordered by pair id on every rank."""
    return gather_rows(np.asarray(local, np.float64).reshape(-1, RECORD), n_pairs, device)
