"""buffer-x_amd/dist.py::spawn_ranks -- the single-node launcher behind `python bench.py --gpus N` -- must FAIL FAST: a rank that dies
(RCCL initialisation on a missing device, an exception before the first collective) ends the job within seconds with a non-zero exit
code and that rank's stderr, instead of leaving the other ranks in a collective until its timeout.  CPU only (gloo where a collective
is needed)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bufferx_amd  # noqa: E402,F401
from bufferx_amd import dist as D  # noqa: E402

OK_WORKER = r'''
import os, sys
print("rank %s of %s on port %s" % (os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_PORT"]))
sys.stderr.write("note from rank %s\n" % os.environ["RANK"])
'''

# rank 1 dies at once; rank 0 waits in a gloo collective that can never complete (default timeout: minutes)
FAIL_WORKER = r'''
import os, sys, datetime
rank = int(os.environ["RANK"])
if rank == 1:
    sys.stderr.write("deliberate failure: no such device\n")
    sys.exit(3)
import torch.distributed as dist
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
dist.barrier()
'''

KILLED_WORKER = r'''
import os, signal, time
if os.environ["RANK"] == "2":
    os.kill(os.getpid(), signal.SIGKILL)
time.sleep(600)
'''


def test_all_ranks_ok(tmp_path):
    w = tmp_path / "ok.py"
    w.write_text(OK_WORKER)
    rc, out0, logs = D.spawn_ranks(3, [sys.executable, str(w)], log_dir=str(tmp_path / "logs"))
    assert rc == 0 and out0.startswith("rank 0 of 3 on port ")
    for r in range(3):      # per-rank logs are kept
        assert open(os.path.join(logs, "rank%d.out" % r)).read().startswith("rank %d of 3" % r)
        assert open(os.path.join(logs, "rank%d.err" % r)).read() == "note from rank %d\n" % r


def test_failing_rank_ends_the_job_fast(tmp_path, capfd):
    w = tmp_path / "fail.py"
    w.write_text(FAIL_WORKER)
    t0 = time.monotonic()
    rc, out0, logs = D.spawn_ranks(2, [sys.executable, str(w)], log_dir=str(tmp_path / "logs"), grace_s=3.0)
    dt = time.monotonic() - t0
    err = capfd.readouterr().err
    assert rc == 3, rc
    assert dt < 60, "launcher took %.1f s: rank 0 was left in its collective" % dt      # (most of it is `import torch` in rank 0)
    assert "rank 1 of 2 exited with code 3" in err and "deliberate failure: no such device" in err


def test_rank_killed_by_a_signal_is_a_failure(tmp_path, capfd):
    w = tmp_path / "killed.py"
    w.write_text(KILLED_WORKER)
    t0 = time.monotonic()
    rc, _, _ = D.spawn_ranks(3, [sys.executable, str(w)], log_dir=str(tmp_path / "logs"), grace_s=2.0)
    assert rc != 0 and time.monotonic() - t0 < 20
    assert "rank 2 of 3 exited with code -9" in capfd.readouterr().err


def test_numa_binding_from_sysfs(tmp_path):
    """bind_rank_to_gpu reads the GPU's NUMA node the way `rocm-smi --showtoponuma` does; falls back to an even split; never raises."""
    allowed = sorted(os.sched_getaffinity(0))
    try:
        pci = "0000:c1:00.0"
        d = tmp_path / "bus/pci/devices" / pci
        d.mkdir(parents=True)
        (d / "numa_node").write_text("1\n")
        n = tmp_path / "devices/system/node/node1"
        n.mkdir(parents=True)
        half = allowed[:max(1, len(allowed) // 2)]
        (n / "cpulist").write_text(",".join(str(c) for c in half) + "\n")
        assert D.gpu_numa_cpus(pci, str(tmp_path)) == set(half)
        got = D.bind_rank_to_gpu(0, 2, pci, str(tmp_path))
        assert got["source"].startswith("numa node") and os.sched_getaffinity(0) == set(half)
        os.sched_setaffinity(0, allowed)
        (d / "numa_node").write_text("-1\n")      # platform does not say: even split over the local ranks
        if len(allowed) >= 2:
            got = D.bind_rank_to_gpu(1, 2, pci, str(tmp_path))
            per = len(allowed) // 2
            assert got["source"] == "even split" and os.sched_getaffinity(0) == set(allowed[per:2 * per])
        os.sched_setaffinity(0, allowed)
        assert D.bind_rank_to_gpu(0, 1, None)["source"].startswith("unchanged")
        assert D._parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    finally:
        os.sched_setaffinity(0, allowed)


def test_bench_gpus2_without_devices_fails_fast():
    """`python bench.py --gpus 2` where the ranks cannot come up (here: no GPU at all) exits non-zero with the failing rank's
    stderr -- what the driver sees instead of a hang when a rank dies at RCCL initialisation."""
    import subprocess
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs (on a GPU box the ranks come up)")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and time.monotonic() - t0 < 120
    assert "[bx launcher] rank" in out.stderr and "exited with code" in out.stderr
