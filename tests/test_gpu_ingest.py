"""Prefetcher (bx_prefetch_*) on the GPU: files -> pinned memory -> async H2D -> device tensors that the hot
path consumes; slot reuse is ordered by events, not by host synchronisation."""
import numpy as np
import pytest

from oracle import io_oracle as IO

pytestmark = pytest.mark.gpu


def test_prefetch_round_trip_and_slot_reuse(tmp_path, bx, packed):
    import torch
    from bufferx_amd import ingest, lib
    rng = np.random.default_rng(0)
    pairs = []
    for i in range(5):
        a = (rng.normal(size=(20000 + 1000 * i, 3)) * 2).astype(np.float32)
        b = (rng.normal(size=(15000 + 500 * i, 3)) * 2).astype(np.float32)
        if i % 3 == 0:
            fa, fb = str(tmp_path / f"a{i}.ply"), str(tmp_path / f"b{i}.ply")
            IO.write_ply(fa, a); IO.write_ply(fb, b, "ascii")
        elif i % 3 == 1:
            fa, fb = str(tmp_path / f"a{i}.pcd"), str(tmp_path / f"b{i}.pcd")
            IO.write_pcd(fa, a, "binary_compressed"); IO.write_pcd(fb, b, "binary", extra=[("intensity", "F4", np.zeros(len(b)))])
        else:
            fa, fb = str(tmp_path / f"a{i}.bin"), str(tmp_path / f"b{i}.bin")
            np.concatenate([a, np.zeros((len(a), 1), np.float32)], 1).tofile(fa)
            np.concatenate([b, np.ones((len(b), 1), np.float32)], 1).tofile(fb)
        pairs.append((fa, fb, a, b))
    pf = ingest.Prefetcher(device=0, slots=2, max_points=30000)
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 64, 64, 1
    cfg.patch.search_radius_thresholds = [5]
    ctx = lib.Context(cfg, max_points=30000, device=0, packed_weights=packed)
    try:
        tickets = [pf.submit(pairs[0][0], pairs[0][1]), pf.submit(pairs[1][0], pairs[1][1])]
        with pytest.raises(lib.BxError):                       # both slots busy
            pf.submit(pairs[2][0], pairs[2][1])
        for i in range(5):
            src, tgt = pf.wait(tickets[i])
            assert src.is_cuda and src.shape == pairs[i][2].shape and tgt.shape == pairs[i][3].shape
            idx, kp = ctx.fps(src, 64)                         # the hot path reads the prefetched buffer on the same stream
            got_s, got_t = src.cpu().numpy(), tgt.cpu().numpy()
            assert np.array_equal(got_s, pairs[i][2]) and np.array_equal(got_t, pairs[i][3])
            assert np.array_equal(kp.cpu().numpy(), pairs[i][2][idx.cpu().numpy()])
            pf.release(tickets[i])
            if i + 2 < 5:
                tickets.append(pf.submit(pairs[i + 2][0], pairs[i + 2][1]))
        bad = pf.submit(str(tmp_path / "nope.ply"), pairs[0][1])
        with pytest.raises(lib.BxError):
            pf.wait(bad)
        t = pf.submit(pairs[0][0], pairs[0][1])                # the failed slot is usable again
        src, _ = pf.wait(t)
        assert np.array_equal(src.cpu().numpy(), pairs[0][2])
        pf.release(t)
    finally:
        ctx.close()
        pf.close()
