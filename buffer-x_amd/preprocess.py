"""Mirror of the reference's per-pair pre-processing on top of the HIP entry points bx_pre_* (SURVEY.md §8f rank 1):

    sphericity_based_voxel_analysis(src_pts, tgt_pts)   <- utils/tools.py:152-198 (+ compute_pca_alignment :132-149)
    voxel_down_sample(pts, voxel_size)                   <- o3d.geometry.PointCloud.voxel_down_sample as called at
                                                            dataset/threedmatch.py:90-102, dataset/kitti.py, dataset/tiers.py

Same names, argument meaning and return values as the reference functions (which take Open3D point clouds; here float32
[n,3] arrays / device tensors).  The random 10 % subsample is drawn with the reference's own call, np.random.choice(num_points,
size=int(num_points / 10), replace=False), in the same order (src, then tgt), so seeding NumPy reproduces it.  There is no CPU
fallback: the PCA statistics, the z-range over the full cloud and the voxel grid run on the GPU."""
import numpy as np

from . import lib as _lib


class Preprocessor:
    def __init__(self, ctx, max_points, upload=None):
        """upload: optional callable (key, numpy array) -> device tensor used for the subsample indices (harness: pinned staging +
        persistent device buffers); the default lets torch allocate."""
        self.ctx = ctx
        self.upload = upload
        ctx.pre_reserve(max_points)
        t = ctx.torch
        self._st = t.empty((2, 17), dtype=t.float64, device=f"cuda:{ctx.device}")       # PCA statistics of src / tgt
        self._cnt = t.empty((8, 2), dtype=t.int32, device=f"cuda:{ctx.device}")          # voxel counts of up to 8 clouds

    def _pca_launch(self, pts_dev, num_points, slot, idx=None):
        if idx is None:
            idx = np.random.choice(num_points, size=int(num_points / 10), replace=False).astype(np.int32)
            if self.upload is not None:
                idx = self.upload(f"pca_idx{slot}", idx)
        return self.ctx.pre_pca(pts_dev, idx, out=self._st[slot])   # device float64 [17]; nothing waits for it yet

    @staticmethod
    def _pca_stats(st):
        ev, comp = st[0:3], st[3:12].reshape(3, 3)
        l1, l2, l3 = sorted(ev, reverse=True)
        sphericity = l3 / l1
        z = comp[-1] / np.linalg.norm(comp[-1])
        is_aligned = abs(float(np.dot(z, np.array([0, 0, 1])))) > 0.98
        return sphericity, is_aligned, comp, float(st[16] - st[15])

    def sphericity_based_voxel_analysis(self, src_pts, tgt_pts, sample_idx=None):
        """-> (voxel_size, sphericity, is_aligned_to_global_z)   (utils/tools.py:152-198).  sample_idx: optional pair of device
        int32 tensors, the 10 % subsamples (default: drawn with the reference's np.random.choice calls)"""
        t = self.ctx.torch
        src = self.ctx._dev(src_pts, t.float32)
        tgt = self.ctx._dev(tgt_pts, t.float32)
        self._pca_launch(src, src.shape[0], 0, None if sample_idx is None else sample_idx[0])
        self._pca_launch(tgt, tgt.shape[0], 1, None if sample_idx is None else sample_idx[1])
        st = self._st.cpu().numpy()                                                     # ONE host sync
        s_src, a_src, c_src, zr_src = self._pca_stats(st[0])
        s_tgt, a_tgt, c_tgt, zr_tgt = self._pca_stats(st[1])
        if src.shape[0] > tgt.shape[0]:
            sphericity, z_range = s_src, zr_src
        else:
            sphericity, z_range = s_tgt, zr_tgt
        alpha = 1.0 if sphericity < 0.05 else 1.5
        voxel_size = max(np.sqrt(z_range) / 100 * alpha, 0.001)
        zs = c_src[-1] / np.linalg.norm(c_src[-1])
        zt = c_tgt[-1] / np.linalg.norm(c_tgt[-1])
        same_direction = float(np.dot(zs, zt)) > 0.96
        return round(float(voxel_size), 4), float(sphericity), bool(a_src and a_tgt and same_direction)

    def voxel_down_sample(self, pts, voxel_size):
        """-> float32 device tensor [m,3] (order of first appearance; the loaders shuffle afterwards)"""
        out, cnt = self.ctx.pre_voxel_downsample(pts, voxel_size)
        m, status = (int(v) for v in cnt.cpu().numpy())
        if status:
            raise _lib.BxError("voxel size too small for the extent of the cloud (more than 2^21 voxels along an axis)")
        return out[:m]

    def voxel_down_sample_many(self, clouds, voxel_size, outs=None):
        """Several (<= 8) clouds, ONE host synchronisation for all the voxel counts -> list of float32 device tensors [m_i,3];
        outs: optional caller-owned output buffers (>= len(cloud) rows each)."""
        res = [self.ctx.pre_voxel_downsample(c, voxel_size, out=None if outs is None else outs[k], cnt=self._cnt[k])
               for k, c in enumerate(clouds)]
        cnt = self._cnt[:len(clouds)].cpu().numpy()
        if cnt[:, 1].any():
            raise _lib.BxError("voxel size too small for the extent of the cloud (more than 2^21 voxels along an axis)")
        return [r[0][:int(m)] for r, m in zip(res, cnt[:, 0])]
