"""Multi-GPU plan: pairs are independent units (reference test.py:132-146 carries no state between pairs), so
ranks take pairs round-robin and the ONLY collective is one all-gather of a fixed-size record per pair at
the end (RCCL over xGMI when backend == "nccl"; gloo on CPU for the tests).  SURVEY.md §8e.

The record is float64 (192 bytes per pair: 3 404 pairs of 3DMatch + 3DLoMatch = 654 KB, still one latency-bound collective): an
un-refined RANSAC pose is binary64 (every outdoor configuration), so a float32 record would change its last bits when it crosses
ranks and a sharded run would not be bit-identical to a single-GPU run.  buffer-x_amd/evaluate.py's state rows are float64 for the
same reason; both travel through gather_rows()."""
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

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


# ---------------------------------------------------------------------------------------------------------------- launcher
def spawn_ranks(n, argv, log_dir=None, poll_s=0.05, grace_s=5.0, extra_env=None):
    """Single-node launcher behind `python bench.py --gpus N` (no torchrun): N ranks of `argv` (one per GPU, LOCAL_RANK = GPU index)
    with the torch.distributed environment of a one-node job (rendezvous on 127.0.0.1).  FAILS FAST: every child is polled; the
    first rank that exits non-zero (e.g. RCCL initialisation on a missing device) ends the job -- the other ranks, which would
    otherwise sit in a collective until its timeout, are terminated (SIGTERM, SIGKILL after `grace_s`), the failing rank's stderr
    tail goes to this process' stderr and its exit code is returned.  stdout / stderr of every rank are kept in
    `log_dir`/rank<r>.{out,err}; rank 0's stdout (the ONE JSON line) is returned.  -> (exit code, rank-0 stdout, log_dir)"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    log_dir = log_dir or tempfile.mkdtemp(prefix="bx_ranks_")
    os.makedirs(log_dir, exist_ok=True)
    procs, files = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BX_RANK_LOG_DIR=log_dir)
        env.update(extra_env or {})
        fo = open(os.path.join(log_dir, "rank%d.out" % r), "wb")
        fe = open(os.path.join(log_dir, "rank%d.err" % r), "wb")
        files += [fo, fe]
        procs.append(subprocess.Popen(list(argv), env=env, stdout=fo, stderr=fe, start_new_session=True))

    def tail(r, nbytes=3000):
        try:
            with open(os.path.join(log_dir, "rank%d.err" % r), "rb") as f:
                f.seek(0, 2)
                f.seek(max(0, f.tell() - nbytes))
                return f.read().decode(errors="replace")
        except OSError:
            return ""

    def stop_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)      # exactly the process groups started above
                except OSError:
                    pass
        t_end = time.monotonic() + grace_s
        for p in procs:
            while p.poll() is None and time.monotonic() < t_end:
                time.sleep(poll_s)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()

    rc, failed = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed, rc = bad[0], codes[bad[0]]
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
    except KeyboardInterrupt:
        failed, rc = -1, 130
    finally:
        if failed is not None:
            stop_all()
        for f in files:
            f.close()
    if failed is not None and failed >= 0:
        sys.stderr.write("[bx launcher] rank %d of %d exited with code %s; the other ranks were stopped.  Logs: %s\n--- rank %d stderr (tail) ---\n%s\n"
                         % (failed, n, rc, log_dir, failed, tail(failed)))
        sys.stderr.flush()
        rc = rc if isinstance(rc, int) and rc > 0 else 1      # a signal death (negative code) is still a failure of the job
    with open(os.path.join(log_dir, "rank0.out"), "rb") as f:
        out0 = f.read().decode(errors="replace")
    return rc, out0, log_dir


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(pci_bus_id, sysfs="/sys"):
    """CPUs of the NUMA node a GPU hangs off: /sys/bus/pci/devices/<dddd:bb:dd.f>/numa_node -> /sys/devices/system/node/node<N>/cpulist
    (the information `rocm-smi --showtoponuma` prints).  None when the platform does not say (numa_node = -1, a VM without the files)."""
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as f:
            return _parse_cpulist(f.read()) or None
    except (OSError, ValueError):
        return None


def bind_rank_to_gpu(local_rank, local_world, pci_bus_id=None, sysfs="/sys"):
    """One process per GPU: keep rank r's host threads (launch loop, harvest, prefetch workers) on the CPUs next to GPU r.
    NUMA node of the GPU when the platform exposes it, else an even split of the allowed CPUs over the local ranks.  Never fatal.
    -> {"cpus": n, "source": "numa node of <pci>" | "even split" | "unchanged (<why>)"}"""
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError) as e:
        return {"cpus": 0, "source": "unchanged (%s)" % e}
    want, source = None, None
    if pci_bus_id:
        cpus = gpu_numa_cpus(pci_bus_id, sysfs)
        if cpus and (cpus & allowed):
            want, source = cpus & allowed, "numa node of %s" % pci_bus_id
    if want is None and local_world > 1 and len(allowed) >= local_world:
        order = sorted(allowed)
        per = len(order) // local_world
        want, source = set(order[local_rank * per:(local_rank + 1) * per]), "even split"
    if not want:
        return {"cpus": len(allowed), "source": "unchanged (one rank or fewer CPUs than ranks)"}
    try:
        os.sched_setaffinity(0, want)
    except OSError as e:
        return {"cpus": len(allowed), "source": "unchanged (%s)" % e}
    return {"cpus": len(want), "source": source}
