"""ctypes view of oracle/liboracle_knn.so (knn_ref.c) -- test infrastructure only."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, 'liboracle_knn.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} missing: run `make -C oracle` (or __graft_entry__.build())')
        _lib = ctypes.CDLL(path)
    return _lib


def knn_packed(ref, query, ref_off, query_off, K):
    ref = np.ascontiguousarray(ref, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    ref_off = np.ascontiguousarray(ref_off, np.int32)
    query_off = np.ascontiguousarray(query_off, np.int32)
    out = np.empty((query.shape[0], K), np.int64)
    _load().oracle_knn_packed(ref.ctypes.data_as(ctypes.c_void_p), query.ctypes.data_as(ctypes.c_void_p),
                              ref_off.ctypes.data_as(ctypes.c_void_p), query_off.ctypes.data_as(ctypes.c_void_p),
                              ctypes.c_int(len(ref_off) - 1), ctypes.c_int(K), out.ctypes.data_as(ctypes.c_void_p))
    return out


def knn_inverse(idx, total_points):
    idx = np.ascontiguousarray(idx, np.int64)
    Nq, K = idx.shape
    inv_n = np.empty(Nq * K, np.int32)
    inv_k = np.empty(Nq * K, np.uint8)
    inv_idx = np.empty(total_points + 1, np.int32)
    _load().oracle_knn_inverse(idx.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(Nq), ctypes.c_int(K),
                               ctypes.c_int(total_points), inv_n.ctypes.data_as(ctypes.c_void_p),
                               inv_k.ctypes.data_as(ctypes.c_void_p), inv_idx.ctypes.data_as(ctypes.c_void_p))
    return inv_n, inv_k, inv_idx
