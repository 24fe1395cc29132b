"""GPU: the drop-in nn.Module (bufferx_amd.model.BufferX) behind the reference harness surface (test.py:83-106,145)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def test_dropin_forward_matches_oracle(bx, oracle):
    from bufferx_amd.model import BufferX
    from oracle import pipeline as PL
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 160, 96, 2
    cfg.patch.search_radius_thresholds = [2, 1]
    cfg.patch.num_points_radius_estimate = 160
    cfg.match.iter_n = 3000
    cfg.test.enable_timing = True
    sd = bx.weights.synthetic_state_dict(0)
    model = BufferX(cfg)
    # reference-style loading (test.py:86-94)
    for stage in ("Desc", "Pose"):
        part = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if stage in k}
        new = model.state_dict()
        new.update(part)
        model.load_state_dict(new)
    model = model.to("cuda:0")
    model = nn.DataParallel(model, device_ids=[0])
    model.eval()
    pair = bx.synth.make_pair(3, "indoor", n_target=4000, identical=True)
    data = {"src_fds_pcd": torch.from_numpy(pair["src"]).cuda(), "tgt_fds_pcd": torch.from_numpy(pair["tgt"]).cuda(),
            "is_aligned_to_global_z": pair["aligned_z"]}
    np.random.seed(123)
    with torch.no_grad():
        pose, times, n_inl, n_mut, n_ind, scales = model(data)
    assert pose.shape == (4, 4) and pose.dtype == np.float32 and len(times) == 3 and times[0] > 0
    # replay the RNG draws the module made, run the oracle with the same permutations / seed
    np.random.seed(123)
    ps, pt = [], []
    for _ in range(2):
        ps.append(np.random.choice(len(pair["src"]), len(pair["src"]), replace=False))
        pt.append(np.random.choice(len(pair["tgt"]), len(pair["tgt"]), replace=False))
    seed = int(np.random.randint(0, 2**31 - 1))
    pw = bx.weights.fold_and_pack(sd)
    orig = oracle.make_perm
    try:
        table = {(2 * i): ps[i] for i in range(2)}
        table.update({(2 * i + 1): pt[i] for i in range(2)})
        oracle.make_perm = lambda n, s, stream: table[stream].astype(np.int32)
        PL.O.make_perm = oracle.make_perm
        ref = PL.register_pair(pair["src"], pair["tgt"], pw, cfg, pair["aligned_z"], seed)
    finally:
        oracle.make_perm = orig
        PL.O.make_perm = orig
    assert (n_inl, n_mut, n_ind, scales) == tuple(ref[1:])
    assert np.array_equal(pose, np.asarray(ref[0], np.float32))
    assert n_mut > 0 and scales == 2


def test_library_is_the_path_that_runs(bx, packed):
    """No silent fallback: the HIP shared object must be loaded, and errors surface as exceptions."""
    from bufferx_amd import lib
    so = lib.load()
    assert so is not None
    maps = open("/proc/self/maps").read()
    assert "libbufferx_hip.so" in maps
    cfg = bx.make_cfg("3DMatch")
    cfg.patch.num_fps, cfg.patch.num_points_per_patch, cfg.patch.num_scales = 64, 32, 1
    cfg.patch.search_radius_thresholds = [2]
    ctx = lib.Context(cfg, max_points=100, device=0, packed_weights=packed)
    with pytest.raises(lib.BxError):   # cloud larger than the context's max_points -> error code, not a crash
        ctx.register_pair(np.zeros((500, 3), np.float32) + 1, np.zeros((500, 3), np.float32) + 1, False,
                          np.zeros((1, 500), np.int32), np.zeros((1, 500), np.int32), 1)
    ctx.close()
