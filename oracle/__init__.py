"""oracle -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; ``reart_amd`` (the product) never does.  The arithmetic lives in the C files next
to this one (each function cites the reference file:line it restates); this module is the numpy
binding.  See oracle/oracle.h for the parity status (which parts are pinned by golden vectors
generated from the reference, and which are UNPINNED third-party contracts).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        if hasattr(_lib, 'oracle_flow_loss'):
            _lib.oracle_flow_loss.restype = ctypes.c_double
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return lib().oracle_num_threads()


# ---- knn.c -------------------------------------------------------------------------------
def knn_points(p1, p2, lengths1=None, lengths2=None, K=1):
    """utils/chamfer.py:140-193 -> (dists f32 [N,P1,K], idx i64 [N,P1,K])."""
    p1, p2 = _f(p1), _f(p2)
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    l1 = None if lengths1 is None else np.ascontiguousarray(lengths1, dtype=np.int64)
    l2 = None if lengths2 is None else np.ascontiguousarray(lengths2, dtype=np.int64)
    dists = np.empty((N, P1, K), np.float32)
    idx = np.empty((N, P1, K), np.int64)
    lib().oracle_knn_points(_p(p1), _p(p2), _p(l1), _p(l2), N, P1, P2, D, K, _p(dists), _p(idx))
    return dists, idx


def knn_points_backward(p1, p2, idx, grad_dists, lengths1=None, lengths2=None):
    """utils/chamfer.py:195-209 -> (grad_p1, grad_p2)."""
    p1, p2, g = _f(p1), _f(p2), _f(grad_dists)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    K = idx.shape[2]
    l1 = None if lengths1 is None else np.ascontiguousarray(lengths1, dtype=np.int64)
    l2 = None if lengths2 is None else np.ascontiguousarray(lengths2, dtype=np.int64)
    g1, g2 = np.empty_like(p1), np.empty_like(p2)
    lib().oracle_knn_points_backward(_p(p1), _p(p2), _p(l1), _p(l2), _p(idx), _p(g), N, P1, P2, D, K, _p(g1), _p(g2))
    return g1, g2


def chamfer_bidir(x, y):
    """utils/chamfer.py:78-123 with bidirectional=True -> (d_xy, i_xy, d_yx, i_yx)."""
    d1, i1 = knn_points(x, y)
    d2, i2 = knn_points(y, x)
    return d1[..., 0], i1[..., 0], d2[..., 0], i2[..., 0]


def knn_cuda(ref, query, k, euclidean=True):
    """knn_cuda.KNN(k, transpose_mode=True)(ref, query) -> (dist [B,nq,k], idx [B,nq,k])."""
    ref, query = _f(ref), _f(query)
    B, nr, D = ref.shape
    nq = query.shape[1]
    dist = np.empty((B, nq, k), np.float32)
    idx = np.empty((B, nq, k), np.int64)
    lib().oracle_knn_cuda(_p(ref), _p(query), B, nr, nq, D, k, int(bool(euclidean)), _p(dist), _p(idx))
    return dist, idx
