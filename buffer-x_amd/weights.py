"""Weight handling for the HIP hot path: state-dict layout, BatchNorm folding, kernel packing, tap tables.

PyTorch is used for weight *loading only* (north_star).  The layer list and key names restate the
reference modules (models/patch_embedder.py:26-41, models/patchnet.py:68-84 Cylindrical_Net,
:192-210 CostNet).  Eval-mode BatchNorm (eps 1e-5) is folded into the conv:
    W' = W * g / sqrt(var + eps),   b' = (b - mean) * g / sqrt(var + eps) + beta
(g = 1, beta = 0 where affine=False), computed in float64 and rounded once to float32.

Kernel weight layout (B operand of the implicit GEMM):  W[chunk][tap][16][cout]
    chunk = group of 16 input channels, tap = kernel offset; accumulation order chunk > tap > c.
"""
import numpy as np

BN_EPS = 1e-5

# (conv key, bn key or None, bn affine, relu)
DESC_CONVS = [
    ("Desc.conv_net.ops.0", "Desc.conv_net.ops.1", False, True),    # Conv3d 16->64 k3^3
    ("Desc.conv_net.ops.3", "Desc.conv_net.ops.4", False, True),    # 64->64
    ("Desc.conv_net.ops.6", "Desc.conv_net.ops.7", False, True),    # 64->128
    ("Desc.conv_net.ops.9", "Desc.conv_net.ops.10", False, True),   # 128->128
    ("Desc.conv_net.ops.12", "Desc.conv_net.ops.13", False, True),  # 128->64
    ("Desc.conv_net.ops.15", "Desc.conv_net.ops.16", False, True),  # 64->64
    ("Desc.conv_net.ops.18", "Desc.conv_net.ops.19", False, True),  # 64->32
    ("Desc.conv_net.ops.21", None, False, False),                   # 32->32 bare
]
POSE_CONVS = [(f"Pose.conv.ops.{3 * i}", f"Pose.conv.ops.{3 * i + 1}" if i < 9 else None, False, i < 9)
              for i in range(10)]

# shapes of every tensor of the reference BufferX state_dict (105 tensors), in registration order
def state_dict_spec():
    spec = []

    def conv(name, shape):
        spec.append((name + ".weight", shape))
        spec.append((name + ".bias", (shape[0],)))

    def bn(name, c, affine):
        if affine:
            spec.append((name + ".weight", (c,)))
            spec.append((name + ".bias", (c,)))
        spec.append((name + ".running_mean", (c,)))
        spec.append((name + ".running_var", (c,)))
        spec.append((name + ".num_batches_tracked", ()))

    conv("Desc.pnt_layer.0", (16, 3, 1, 1)); bn("Desc.pnt_layer.1", 16, True)
    conv("Desc.pool_layer.0", (16, 32, 1, 1)); bn("Desc.pool_layer.1", 16, True)
    conv("Desc.pool_layer.3", (1, 16, 1, 1)); bn("Desc.pool_layer.4", 1, True)
    conv("Desc.conv_net.ops.0", (64, 16, 3, 3, 3)); bn("Desc.conv_net.ops.1", 64, False)
    chans = [(64, 64), (64, 128), (128, 128), (128, 64), (64, 64), (64, 32)]
    for i, (ci, co) in enumerate(chans):
        conv(f"Desc.conv_net.ops.{3 + 3 * i}", (co, ci, 3, 3)); bn(f"Desc.conv_net.ops.{4 + 3 * i}", co, False)
    conv("Desc.conv_net.ops.21", (32, 32, 3, 3))
    pose = [(32, 32, (3, 3, 3)), (32, 64, (3, 3, 3)), (64, 64, (3, 1, 3)), (64, 128, (3, 1, 3)), (128, 128, (3, 1, 3)),
            (128, 64, (3, 1, 3)), (64, 64, (3, 1, 3)), (64, 32, (3, 1, 3)), (32, 32, (3, 1, 3))]
    for i, (ci, co, k) in enumerate(pose):
        conv(f"Pose.conv.ops.{3 * i}", (co, ci) + k); bn(f"Pose.conv.ops.{3 * i + 1}", co, False)
    conv("Pose.conv.ops.27", (20, 32, 2, 1, 2))
    return spec


def synthetic_state_dict(seed=0):
    """Seeded random weights in the reference snapshot layout (no checkpoints exist offline).
    He-style uniform conv weights keep activations O(1); BN running stats are randomised so that the
    folding is exercised."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in state_dict_spec():
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = np.array(100, np.int64)
        elif leaf == "running_mean":
            sd[name] = rng.normal(0, 0.1, shape).astype(np.float32)
        elif leaf == "running_var":
            sd[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1 and leaf == "weight":      # BN gamma
            sd[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1:                            # conv bias / BN beta
            sd[name] = rng.normal(0, 0.1, shape).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            b = np.sqrt(6.0 / fan_in)
            sd[name] = rng.uniform(-b, b, shape).astype(np.float32)
    return sd


def _fold(sd, conv, bn, affine):
    W = np.asarray(sd[conv + ".weight"], np.float64)
    b = np.asarray(sd[conv + ".bias"], np.float64)
    if bn is not None:
        mean = np.asarray(sd[bn + ".running_mean"], np.float64)
        var = np.asarray(sd[bn + ".running_var"], np.float64)
        g = np.asarray(sd[bn + ".weight"], np.float64) if affine else np.ones_like(mean)
        beta = np.asarray(sd[bn + ".bias"], np.float64) if affine else np.zeros_like(mean)
        s = g / np.sqrt(var + BN_EPS)
        W = W * s.reshape((-1,) + (1,) * (W.ndim - 1))
        b = (b - mean) * s + beta
    return W.astype(np.float32), b.astype(np.float32)


def _pack_chunked(W):
    """W [cout][cin][taps] -> [cin/16][taps][16][cout]"""
    cout, cin, taps = W.shape
    assert cin % 16 == 0
    return np.ascontiguousarray(W.reshape(cout, cin // 16, 16, taps).transpose(1, 3, 2, 0))


def fold_and_pack(sd):
    """state_dict (numpy or torch tensors) -> dict of float32 arrays in kernel layout."""
    sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}
    out = {}
    W, b = _fold(sd, "Desc.pnt_layer.0", "Desc.pnt_layer.1", True)
    out["pnt_w"], out["pnt_b"] = W.reshape(16, 3).copy(), b
    W, b = _fold(sd, "Desc.pool_layer.0", "Desc.pool_layer.1", True)
    out["pool_w1"], out["pool_b1"] = W.reshape(16, 32).copy(), b
    W, b = _fold(sd, "Desc.pool_layer.3", "Desc.pool_layer.4", True)
    out["pool_w2"], out["pool_b2"] = W.reshape(16).copy(), b.reshape(1)
    desc = []
    for i, (conv, bn, aff, relu) in enumerate(DESC_CONVS):
        W, b = _fold(sd, conv, bn, aff)
        if i == 0:
            # Conv3d [co][ci=16][kd][kh][kw]: chunk = kd (radial shell), tap = kh*3+kw, c = ci
            Wk = np.ascontiguousarray(W.transpose(2, 3, 4, 1, 0).reshape(3, 9, 16, W.shape[0]))
        else:
            Wk = _pack_chunked(W.reshape(W.shape[0], W.shape[1], 9))
        desc.append(dict(W=Wk, b=b, relu=relu, cout=W.shape[0]))
    out["desc"] = desc
    pose = []
    for i, (conv, bn, aff, relu) in enumerate(POSE_CONVS):
        W, b = _fold(sd, conv, bn, aff)
        k = W.shape[2:]
        Wk = _pack_chunked(W.reshape(W.shape[0], W.shape[1], int(np.prod(k))))
        pose.append(dict(W=Wk, b=b, relu=relu, cout=W.shape[0], k=tuple(int(v) for v in k)))
    out["pose"] = pose
    return out


# ---------------------------------------------------------------- tap tables
def cyl_tap_table(ele_n=7, azi_n=20):
    """3x3 taps on the (elevation, azimuth) map: circular in azimuth, zero in elevation
    (utils/common.py:265-310 pad_image / pad_image_3d)."""
    t = np.full((9, ele_n * azi_n), -1, np.int32)
    for kh in range(3):
        for kw in range(3):
            for h in range(ele_n):
                hh = h + kh - 1
                if hh < 0 or hh >= ele_n:
                    continue
                for w in range(azi_n):
                    t[kh * 3 + kw, h * azi_n + w] = hh * azi_n + (w + kw - 1) % azi_n
    return t


def valid_tap_table(dims, k):
    """Un-padded ("valid") 3-D convolution taps, CostNet (models/patchnet.py:192-210)."""
    D, H, W = dims
    kd, kh, kw = k
    Do, Ho, Wo = D - kd + 1, H - kh + 1, W - kw + 1
    t = np.zeros((kd * kh * kw, Do * Ho * Wo), np.int32)
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                for d in range(Do):
                    for h in range(Ho):
                        for w in range(Wo):
                            t[(a * kh + b) * kw + c, (d * Ho + h) * Wo + w] = ((d + a) * H + (h + b)) * W + (w + c)
    return t, (Do, Ho, Wo)


def pose_geometry(ele_n=7, azi_n=20):
    """Per-layer (in_dims, kernel, out_dims) of CostNet on the [azi, ele-2, azi] cost volume."""
    dims = (azi_n, ele_n - 2, azi_n)
    ks = [(3, 3, 3), (3, 3, 3)] + [(3, 1, 3)] * 7 + [(2, 1, 2)]
    geo = []
    for k in ks:
        out = (dims[0] - k[0] + 1, dims[1] - k[1] + 1, dims[2] - k[2] + 1)
        geo.append((dims, k, out))
        dims = out
    return geo
