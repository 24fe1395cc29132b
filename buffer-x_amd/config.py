"""Hot-path configuration knobs, mirroring the reference's EasyDict trees.

Reference: config/indoor_config.py:4-80, config/outdoor_config.py:4-82 and the per-dataset
subclasses (config/threedmatch_config.py:12 sets test.pose_refine=True, ...).  Only the knobs the
inference path reads are carried; CLI overrides follow test.py:51-78.
"""


class Cfg(dict):
    """Minimal attribute-dict with EasyDict semantics (cfg.a.b, cfg.a.get('b', default))."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Cfg):
            v = Cfg(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


# Arithmetic forms of the three stages that have more than one (include/bufferx.h bx_params.desc_conv_form / pose_conv_form /
# cost_l0_form; no reference counterpart: the reference leaves the summation order of its convolutions to cuDNN / ATen).  Index in the
# tuple = the C-ABI value; the first entry of each is the default.  cfg.arith carries the names; the oracle takes the same names.
ARITH_FORMS = {"desc_conv": ("winograd43", "winograd22", "direct", "winograd43m"), "pose_conv": ("winograd43", "winograd22", "direct"), "cost_l0": ("collapsed", "direct")}
# what make_cfg puts into cfg.arith.  tests/conftest.py --arith overrides it for a whole test run (every form is covered that way);
# nothing reads the process environment.
ARITH_DEFAULT = {k: v[0] for k, v in ARITH_FORMS.items()}


def arith_of(cfg):
    """The validated arithmetic forms of a knob tree (missing keys = defaults)."""
    ar = dict(ARITH_DEFAULT)
    ar.update(dict(cfg.get("arith", {}) or {}))
    for k, v in ar.items():
        if k not in ARITH_FORMS or v not in ARITH_FORMS[k]:
            raise ValueError(f"unknown arithmetic form {k}={v!r} (known: {ARITH_FORMS})")
    return ar


def _base(indoor):
    c = Cfg()
    c.arith = dict(ARITH_DEFAULT)
    c.stage = "test"
    c.data = dict(dataset="", voxel_size_0=0.035 if indoor else 0.30)
    c.test = dict(pose_refine=False, enable_timing=False,
                  rte_thresh=0.3 if indoor else 2.0, rre_thresh=15.0 if indoor else 5.0)
    c.patch = dict(des_r=0.3 if indoor else 3.0, num_points_per_patch=512, num_fps=1500, rad_n=3, azi_n=20, ele_n=7,
                   delta=0.8, voxel_sample=10, num_scales=3, is_aligned_to_global_z=not indoor,
                   search_radius_thresholds=[5, 2, 0.5], num_points_radius_estimate=2000)
    if indoor:
        c.match = dict(pose_estimator="ransac", dist_th=0.10, inlier_th=1 / 3, similar_th=0.8, confidence=0.999,
                       iter_n=50000, kiss_resolution=0.3, enable_early_exit=False, early_exit_min_inliers=50)
    else:
        c.match = dict(pose_estimator="ransac", dist_th=0.30, inlier_th=2.0, similar_th=0.9, confidence=1.0,
                       iter_n=50000, kiss_resolution=0.3, enable_early_exit=False, early_exit_min_inliers=50)
    return c


_INDOOR = {"3DMatch", "3DLoMatch", "Scannetpp_iphone", "Scannetpp_faro", "TIERS", "TIERS_hetero"}
_OUTDOOR = {"KITTI", "WOD", "MIT", "KAIST", "KAIST_hetero", "ETH", "Oxford"}


def make_cfg(dataset="3DMatch"):
    """Knob tree for a dataset name (reference config/__init__.py:18-56)."""
    if dataset in ("3DMatch", "3DLoMatch"):
        c = _base(True)
        c.test.pose_refine = True  # config/threedmatch_config.py:12
    elif dataset in ("TIERS", "TIERS_hetero"):
        # reference: config/tiers_config.py derives from OutdoorBaseConfig (is_aligned_to_global_z=True)
        c = _base(False)
    elif dataset in _INDOOR or dataset == "ModelNet40":
        c = _base(True)
        if dataset == "ModelNet40":  # config/modelnet40_config.py:11-12
            c.test.rte_thresh = 0.1
    elif dataset in _OUTDOOR:
        c = _base(False)
        if dataset == "ETH":  # config/eth_config.py:12-19
            c.match.dist_th = 0.20
            c.match.inlier_th = 1.5
            c.test.rte_thresh = 0.3
            c.test.rre_thresh = 2.0
    else:
        raise ValueError(f"unknown dataset {dataset}")
    c.data.dataset = dataset
    return c
