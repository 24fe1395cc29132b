"""world_size-2 gloo test of the multi-GPU plan: round-robin pair sharding + ONE all-gather of fixed-size float64 records."""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BX_ROOT"])
import bufferx_amd
from bufferx_amd import dist as D
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_pairs = 7
mine = D.shard_indices(n_pairs, rank, world)
recs = []
for i in mine:
    T = np.eye(4); T[:3, 3] = [i, 2 * i, 3 * i]; T[0, 1] = 0.5 * i + 1e-13 / 3
    recs.append(D.pack_record(i, T, 10 + i, 100 + i, 50 + i, 3, 1.5 * i))
allr = D.gather_records(np.stack(recs) if recs else np.zeros((0, D.RECORD), np.float64), n_pairs)
assert allr.shape == (n_pairs, D.RECORD), allr.shape
for i in range(n_pairs):
    u = D.unpack_record(allr[i])
    assert u["pair_id"] == i and u["num_inliers"] == 10 + i and u["num_mutual"] == 100 + i
    assert np.array_equal(u["pose"][:3, 3], [i, 2 * i, 3 * i]) and u["scales_used"] == 3
    assert u["pose"].dtype == np.float64 and u["pose"][0, 1] == 0.5 * i + 1e-13 / 3      # binary64 survives the collective
sys.stdout.write(f"rank{rank}ok{len(mine)}\n"); sys.stdout.flush()
dist.destroy_process_group()
'''


def test_shard_and_allgather_world2(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, BX_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(w)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank0ok4" in out.stdout and "rank1ok3" in out.stdout


def test_shard_covers_all_pairs_once():
    from bufferx_amd import dist as D
    for n in (0, 1, 7, 1623):
        for w in (1, 2, 8):
            got = sorted(i for r in range(w) for i in D.shard_indices(n, r, w))
            assert got == list(range(n))
    r = D.pack_record(5, np.eye(4), 1, 2, 3, 4, 6.5, 77)
    u = D.unpack_record(r)
    assert r.nbytes == 8 * D.RECORD and u["model_ms"] == 6.5 and u["ransac_iters"] == 77 and u["pose"].dtype == np.float64
    u32 = D.unpack_record(D.pack_record(5, np.eye(4, dtype=np.float32), 1, 2, 3, 4, 6.5))
    assert u32["pose"].dtype == np.float32
