"""bufferx_amd -- MI355X-native BUFFER-X inference hot path (HIP kernels behind a C-ABI).

Layout: csrc/ (HIP kernels + C-ABI, built into csrc/libbufferx_hip.so), lib.py (ctypes binding),
model.py (drop-in `BufferX` mirroring reference models/BUFFERX.py), weights.py, config.py, synth.py.
"""
from .config import make_cfg, Cfg  # noqa: F401
from . import config, weights, synth  # noqa: F401

__all__ = ["make_cfg", "Cfg", "config", "weights", "synth"]
