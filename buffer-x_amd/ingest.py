"""Data ingest in front of the hot path on top of the native entry points bx_io_* / bx_prefetch_*:

    read_point_cloud(path) -> float32 [n,3]   <- np.asarray(o3d.io.read_point_cloud(path).points) (dataset/threedmatch.py:75-79,
                                                 dataset/tiers.py:72-75) and np.fromfile(path, np.float32).reshape(-1, 4)[:, :3]
                                                 (dataset/kitti.py:76-80), chosen by the extension (.ply / .pcd / .bin)
    Prefetcher                                 a native worker thread reads the NEXT pairs into pinned memory and uploads them on
                                                 its own HIP stream while the GPU registers the current pair; the reference reads
                                                 and uploads every pair synchronously on the main thread (num_workers = 0).

The parsers are native C++ (buffer-x_amd/csrc/k_io.hip); there is no Python fallback."""
import ctypes as C

import numpy as np

from . import lib as _lib


def probe(path):
    n = C.c_int64()
    _lib._chk(_lib.load().bx_io_probe(str(path).encode(), C.byref(n)), "bx_io_probe")
    return int(n.value)


def read_point_cloud(path):
    n = probe(path)
    out = np.empty((max(n, 1), 3), np.float32)
    got = C.c_int64()
    _lib._chk(_lib.load().bx_io_read_xyz(str(path).encode(), out.ctypes.data_as(C.c_void_p), C.c_int64(n), C.byref(got)), "bx_io_read_xyz")
    return out[:int(got.value)]


class Prefetcher:
    """p = Prefetcher(device, slots, max_points); t = p.submit(src_path, tgt_path); src, tgt = p.wait(t)  (float32 [n,3] CUDA
    tensors that alias the prefetcher's device buffers, ordered on the current stream); p.release(t) when the pair has been queued."""

    def __init__(self, device=0, slots=4, max_points=400000):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.device = int(device)
        self.handle = C.c_void_p()
        _lib._chk(self.lib.bx_prefetch_create(C.c_int32(self.device), C.c_int32(int(slots)), C.c_int64(int(max_points)), C.byref(self.handle)),
                  "bx_prefetch_create")

    def submit(self, src_path, tgt_path):
        t = C.c_int64()
        _lib._chk(self.lib.bx_prefetch_submit(self.handle, str(src_path).encode(), str(tgt_path).encode(), C.byref(t)), "bx_prefetch_submit")
        return int(t.value)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def wait(self, ticket):
        ps, pt = C.c_void_p(), C.c_void_p()
        ns, nt = C.c_int64(), C.c_int64()
        _lib._chk(self.lib.bx_prefetch_wait(self.handle, C.c_int64(ticket), self._stream(), C.byref(ps), C.byref(ns), C.byref(pt), C.byref(nt)),
                  "bx_prefetch_wait")
        return self._view(ps.value, ns.value), self._view(pt.value, nt.value)

    def _view(self, ptr, n):
        # zero-copy view of the prefetcher's device buffer through the CUDA array interface
        class _Buf:
            pass
        b = _Buf()
        b.__cuda_array_interface__ = {"shape": (int(n), 3), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        if n == 0:
            return self.torch.empty((0, 3), dtype=self.torch.float32, device=f"cuda:{self.device}")
        return self.torch.as_tensor(b, device=f"cuda:{self.device}")

    def release(self, ticket):
        _lib._chk(self.lib.bx_prefetch_release(self.handle, C.c_int64(ticket), self._stream()), "bx_prefetch_release")

    def close(self):
        if self.handle:
            self.lib.bx_prefetch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
