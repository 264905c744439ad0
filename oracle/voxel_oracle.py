"""ctypes binding of oracle/voxel_oracle.c (the plain-C CPU restatement of the voxel histogram).

TEST INFRASTRUCTURE ONLY -- see the header of voxel_oracle.c. Build with `make -C oracle`
(__graft_entry__.build() does that)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvoxel_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "voxel_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libvoxel_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.evp_oracle_voxel_f32.restype = ctypes.c_int64
        _lib.evp_oracle_voxel_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.evp_oracle_voxel_batch_f32.restype = ctypes.c_int64
        _lib.evp_oracle_voxel_batch_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return _lib


def voxel_grid(events: np.ndarray, bins: int, size, is_txyp=False) -> np.ndarray:
    """events float64 [N,4] -> float32 [bins,H,W]."""
    ev = np.ascontiguousarray(events, dtype=np.float64)
    assert ev.ndim == 2 and ev.shape[1] == 4
    H, W = int(size[0]), int(size[1])
    out = np.empty((bins, H, W), dtype=np.float32)
    _load().evp_oracle_voxel_f32(ev.ctypes.data, ev.shape[0], bins, H, W, int(is_txyp), out.ctypes.data)
    return out


def voxel_grid_batch(events: np.ndarray, offsets: np.ndarray, bins: int, size, is_txyp=False) -> np.ndarray:
    ev = np.ascontiguousarray(events, dtype=np.float64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    H, W = int(size[0]), int(size[1])
    n = off.shape[0] - 1
    out = np.empty((n, bins, H, W), dtype=np.float32)
    _load().evp_oracle_voxel_batch_f32(ev.ctypes.data, off.ctypes.data, n, bins, H, W, int(is_txyp), out.ctypes.data)
    return out
