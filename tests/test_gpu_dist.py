"""Multi-GPU plan on the one GPU of the test box (reference loop test.py:132-192 is one process):
  * the RCCL path itself -- init_process_group("nccl", device_id=...) + ONE all_gather_into_tensor of a float64 DEVICE tensor --
    executed with a world of one (BX_DIST_FORCE_COLLECTIVE keeps dist.gather_rows from short-cutting it);
  * the file-driven loop (harness.Runner, the mirror of test.py:132-146) sharded over two ranks (gloo, both on the one GPU):
    pairs i mod 2 == rank, ONE all-gather of the float64 state rows, rank 0 holds the rows of a single-rank run bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NCCL_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BX_ROOT"])
from bufferx_amd import dist as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
rng = np.random.default_rng(0)
recs = []
for i in (3, 0, 4, 1, 2):
    T = np.eye(4); T[:3, :] = rng.standard_normal((3, 4)); T[0, 1] += 1e-13 / 3
    recs.append(D.pack_record(i, T, 10 + i, 100 + i, 50 + i, 3, 1.5 * i, 7))
loc = np.stack(recs)
out = D.gather_records(loc, 5, device="cuda:0")          # all_gather_into_tensor on a float64 device tensor over RCCL
assert out.dtype == np.float64 and out.shape == loc.shape
assert np.array_equal(out, loc[np.argsort(loc[:, 0])]), "binary64 rows must survive the collective bit for bit"
t = torch.tensor([2.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # bench.py's max-over-ranks timing reduction
dist.barrier()
assert float(t.item()) == 2.5
dist.destroy_process_group()
print("NCCL_WORLD1_OK")
'''


def test_rccl_collective_world1(tmp_path):
    w = tmp_path / "nccl_worker.py"
    w.write_text(NCCL_WORKER)
    env = dict(os.environ, BX_ROOT=ROOT, BX_DIST_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, str(w)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "NCCL_WORLD1_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


RUNNER_WORKER = r'''
import os, sys, json, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BX_ROOT"])
import bufferx_amd as bx
from bufferx_amd import harness, evaluate
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo")
pairs = json.load(open(os.environ["BX_PAIRS"]))
for p in pairs:
    p["relt_pose"] = np.asarray(p["relt_pose"], np.float64)
cfg = bx.make_cfg("3DMatch")
cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 256, 128, 2
cfg.patch.search_radius_thresholds = [5, 2]
cfg.patch.num_points_radius_estimate = 256
pw = bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))
run = harness.Runner(cfg, pw, device=0, inflight=2, max_raw_points=40000, max_points=40000, rng=os.environ["BX_RNG"])
try:
    np.random.seed(5)
    rows, poses = run.run(pairs, rank=rank, world=world, pair_seed=(123 if os.environ["BX_RNG"] == "reference" else None))
finally:
    run.close()
assert [int(r) for r in rows[:, 0]] == list(range(rank, len(pairs), world))
allrows = evaluate.gather_states(rows, len(pairs))
if rank == 0:
    np.save(os.environ["BX_OUT"], allrows)
if world > 1:
    dist.destroy_process_group()
print("RUNNER_OK", rank)
'''


@pytest.mark.parametrize("rng", ["reference", "device"])
def test_runner_sharded_equals_single_rank(tmp_path, bx, rng):
    import json
    from oracle import io_oracle as IO
    r = np.random.default_rng(3)
    pairs = []
    for i in range(5):
        p = bx.synth.make_pair(40 + i, "indoor", n_target=8000, jitter=0.0)
        raw = [np.concatenate([c + r.normal(0, 0.002, c.shape) for _ in range(2)]).astype(np.float32) for c in (p["src"], p["tgt"])]
        fs, ft = str(tmp_path / f"s{i}.ply"), str(tmp_path / f"t{i}.ply")
        IO.write_ply(fs, raw[0]); IO.write_ply(ft, raw[1])
        pairs.append(dict(src_path=fs, tgt_path=ft, relt_pose=p["T_gt"].tolist()))
    (tmp_path / "pairs.json").write_text(json.dumps(pairs))
    w = tmp_path / "runner_worker.py"
    w.write_text(RUNNER_WORKER)
    base = dict(os.environ, BX_ROOT=ROOT, BX_PAIRS=str(tmp_path / "pairs.json"), BX_RNG=rng)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        base.pop(k, None)
    one = subprocess.run([sys.executable, str(w)], env=dict(base, BX_OUT=str(tmp_path / "w1.npy")), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29549", str(w)], env=dict(base, BX_OUT=str(tmp_path / "w2.npy")), capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    a, b = np.load(tmp_path / "w1.npy"), np.load(tmp_path / "w2.npy")
    assert a.shape == b.shape == (5, 32)
    keep = [c for c in range(32) if not 8 <= c <= 12]        # columns 8..12 are wall-clock times
    assert np.array_equal(a[:, keep], b[:, keep])
