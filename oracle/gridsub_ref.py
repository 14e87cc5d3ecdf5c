"""TEST INFRASTRUCTURE -- ctypes view of oracle/_ref/libgridsub_ref.so: the reference's own grid_subsampling()
(compiled by oracle/Makefile from /root/reference/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp).
Used to pin oracle/grid_subsample_oracle.py, to make tests/golden/gridsub_*.npz and as bench.py's
cpu_baseline (kind "reference") of the subsampling workload."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libgridsub_ref.so')


def available():
    return os.path.exists(_PATH)


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
        _lib.gridsub_ref.restype = ctypes.c_int
        _lib.gridsub_ref.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1):
    """-> (sub_points, sub_features or None, sub_labels or None) in the reference's own (hash-map) order."""
    lib = _load()
    pts = np.ascontiguousarray(points, np.float32)
    n = pts.shape[0]
    f = None if features is None else np.ascontiguousarray(features, np.float32)
    c = None if labels is None else np.ascontiguousarray(labels, np.int32).reshape(n, -1)
    fd = 0 if f is None else f.shape[1]
    ld = 0 if c is None else c.shape[1]
    op = np.empty((n, 3), np.float32)
    of = np.empty((n, max(fd, 1)), np.float32)
    oc = np.empty((n, max(ld, 1)), np.int32)
    m = lib.gridsub_ref(pts.ctypes.data, None if f is None else f.ctypes.data, None if c is None else c.ctypes.data, n, fd,
                        ld, float(sampleDl), op.ctypes.data, of.ctypes.data, oc.ctypes.data)
    of = of.reshape(-1)[:m * fd].reshape(m, fd) if fd else None
    oc = oc.reshape(-1)[:m * ld].reshape(m, ld) if ld else None
    return op[:m].copy(), of, oc
