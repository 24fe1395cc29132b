"""tests/golden/make_golden_pre.py -- mints tests/golden/pre_*.npz for the pre-processing row by running
the REAL reference function utils/tools.py::sphericity_based_voxel_analysis (and scikit-learn's PCA) in this container.
Open3D and nibabel are import-time dependencies of utils/tools.py only; they are stubbed (the function touches nothing but
`pcd.points`).  The np.random.choice subsamples are reproduced by seeding NumPy and replaying the same calls.
Run: python tests/golden/make_golden_pre.py   (needs /root/reference; the fixtures are committed)."""
import os
import sys
import types
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


class Pcd:
    def __init__(self, pts):
        self.points = np.asarray(pts, np.float64)


def main():
    for name in ("open3d", "nibabel", "nibabel.quaternions"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nibabel"].quaternions = sys.modules["nibabel.quaternions"]
    sys.path.insert(0, REF)
    from utils import tools as T           # the reference module, unmodified
    import bufferx_amd as bx
    cases = {"pre_indoor": ("indoor", 7, dict(n_target=20000)), "pre_outdoor": ("outdoor", 3, dict(voxel=0.1)),
             "pre_flat": ("indoor", 11, dict(n_target=6000))}
    for name, (kind, seed, kw) in cases.items():
        pair = bx.synth.make_pair(seed, kind, **kw)
        src, tgt = pair["src"], pair["tgt"]
        if name == "pre_flat":              # nearly planar clouds: sphericity < 0.05 branch (alpha = 1.0)
            src = src.copy(); tgt = tgt.copy()
            src[:, 2] *= 0.02; tgt[:, 2] *= 0.02
        np.random.seed(seed)
        vs, sph, aligned = T.sphericity_based_voxel_analysis(Pcd(src), Pcd(tgt))
        np.random.seed(seed)                # replay the two subsample draws (utils/tools.py:136, order src then tgt)
        idx_s = np.random.choice(len(src), size=int(len(src) / 10), replace=False)
        idx_t = np.random.choice(len(tgt), size=int(len(tgt) / 10), replace=False)
        from sklearn.decomposition import PCA
        p = PCA(n_components=3).fit(np.asarray(src, np.float64)[idx_s])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), src=src.astype(np.float32), tgt=tgt.astype(np.float32),
                            idx_src=idx_s.astype(np.int32), idx_tgt=idx_t.astype(np.int32), voxel_size=vs, sphericity=sph,
                            aligned=aligned, ev_src=p.explained_variance_, comp_src=p.components_, mean_src=p.mean_)
        print(name, len(src), len(tgt), vs, sph, aligned)


if __name__ == "__main__":
    main()
