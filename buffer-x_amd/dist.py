"""Multi-GPU plan: pairs are independent units (reference test.py:132-146 carries no state between pairs), so
ranks take pairs round-robin and the ONLY collective is one all-gather of a fixed 72-byte record per pair at
the end (RCCL over xGMI when backend == "nccl"; gloo on CPU for the tests).  SURVEY.md §8e."""
import numpy as np

RECORD = 18  # float32: pair_id, R[9], t[3], num_inliers, num_mutual, num_inlier_ind, scales_used, model_ms  (72 B)


def shard_indices(n_pairs, rank, world):
    """rank r handles pairs {i : i mod W == r}"""
    return list(range(rank, n_pairs, world))


def pack_record(pair_id, pose, num_inliers, num_mutual, num_inlier_ind, scales_used, model_ms):
    r = np.zeros(RECORD, np.float32)
    pose = np.asarray(pose, np.float64).reshape(4, 4)
    r[0] = pair_id
    r[1:10] = pose[:3, :3].reshape(-1)
    r[10:13] = pose[:3, 3]
    r[13:17] = [num_inliers, num_mutual, num_inlier_ind, scales_used]
    r[17] = model_ms
    return r


def unpack_record(r):
    T = np.eye(4)
    T[:3, :3] = np.asarray(r[1:10], np.float64).reshape(3, 3)
    T[:3, 3] = r[10:13]
    return dict(pair_id=int(r[0]), pose=T, num_inliers=int(r[13]), num_mutual=int(r[14]), num_inlier_ind=int(r[15]),
                scales_used=int(r[16]), model_ms=float(r[17]))


def gather_rows(local, n_rows, device=None):
    """local: [n_local, W] (float32 or float64) rows of this rank, unit id in column 0 -> [n_rows, W] ordered by id on every
    rank.  ONE all_gather of equal-sized, -1-padded blocks (RCCL over xGMI with backend nccl; gloo on CPU)."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local)
    assert local.ndim == 2 and local.dtype in (np.float32, np.float64)
    W = local.shape[1]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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
    """local: float32 [n_local, RECORD] of this rank's pairs -> float32 [n_pairs, RECORD] ordered by pair id on every rank."""
    return gather_rows(np.asarray(local, np.float32).reshape(-1, RECORD), n_pairs, device)
