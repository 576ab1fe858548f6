"""Drop-in for the ``chamferdist._C`` extension module the reference imports at
``utils/chamfer.py:12`` and calls at ``utils/chamfer.py:174`` / ``:206-208``.

Same two functions, same argument order, same return order; backed by
``reart_knn_points_idx`` / ``reart_knn_points_backward`` in libreart_hip.so.
"""
import torch

from . import _lib


def _as_len(lengths, n, full, device):
    if lengths is None:
        return None
    if lengths.dtype != torch.int64:
        lengths = lengths.long()
    # a full-length vector is the reference's default (utils/chamfer.py:266-275)
    return lengths.contiguous()


def knn_points_idx(p1, p2, lengths1, lengths2, K, version=-1):
    """-> (idx int64 [N,P1,K], dists float32 [N,P1,K] squared L2), cf. utils/chamfer.py:174."""
    _lib.require_gpu(p1, p2, lengths1, lengths2)
    if p1.dtype != torch.float32 or p2.dtype != torch.float32:
        raise TypeError("knn_points_idx expects float32 point clouds")
    p1, p2 = p1.contiguous(), p2.contiguous()
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    l1 = _as_len(lengths1, N, P1, p1.device)
    l2 = _as_len(lengths2, N, P2, p1.device)
    dists = torch.empty((N, P1, K), dtype=torch.float32, device=p1.device)
    idx = torch.empty((N, P1, K), dtype=torch.int64, device=p1.device)
    L = _lib.lib()
    nbytes = L.reart_knn_points_workspace_bytes(N, P1, P2, K)
    ws = _lib.workspace(nbytes, p1.device)
    rc = L.reart_knn_points_idx(_lib.ptr(p1), _lib.ptr(p2), _lib.ptr(l1), _lib.ptr(l2), N, P1, P2, D, K,
                                _lib.ptr(dists), _lib.ptr(idx), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_knn_points_idx")
    return idx, dists


def knn_points_backward(p1, p2, lengths1, lengths2, idx, grad_dists):
    """-> (grad_p1, grad_p2), cf. utils/chamfer.py:206-208."""
    _lib.require_gpu(p1, p2, idx, grad_dists)
    p1, p2 = p1.contiguous(), p2.contiguous()
    idx, grad_dists = idx.contiguous(), grad_dists.contiguous()
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    K = idx.shape[2]
    l1 = _as_len(lengths1, N, P1, p1.device)
    l2 = _as_len(lengths2, N, P2, p1.device)
    g1 = torch.empty_like(p1)
    g2 = torch.empty_like(p2)
    L = _lib.lib()
    nbytes = L.reart_knn_points_backward_workspace_bytes(N, P1, P2, K)
    ws = _lib.workspace(nbytes, p1.device)
    rc = L.reart_knn_points_backward(_lib.ptr(p1), _lib.ptr(p2), _lib.ptr(l1), _lib.ptr(l2), _lib.ptr(idx),
                                     _lib.ptr(grad_dists), N, P1, P2, D, K, _lib.ptr(g1), _lib.ptr(g2),
                                     _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_knn_points_backward")
    return g1, g2


def chamfer_bidir(x, y):
    """Fused two-direction K=1 search: -> (d_xy, i_xy, d_yx, i_yx), each [N,P]."""
    _lib.require_gpu(x, y)
    x, y = x.contiguous(), y.contiguous()
    N, Pn, _ = x.shape
    d_xy = torch.empty((N, Pn), dtype=torch.float32, device=x.device)
    d_yx = torch.empty_like(d_xy)
    i_xy = torch.empty((N, Pn), dtype=torch.int64, device=x.device)
    i_yx = torch.empty_like(i_xy)
    L = _lib.lib()
    ws = _lib.workspace(L.reart_chamfer_bidir_workspace_bytes(N, Pn), x.device)
    rc = L.reart_chamfer_bidir(_lib.ptr(x), _lib.ptr(y), N, Pn, _lib.ptr(d_xy), _lib.ptr(i_xy), _lib.ptr(d_yx),
                               _lib.ptr(i_yx), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_chamfer_bidir")
    return d_xy, i_xy, d_yx, i_yx


def knn_points_idx_warm(p1, p2, K, seed=None):
    """Warm-started exact search (``reart_knn_points_idx_warm``): same (idx, dists) as
    ``knn_points_idx(p1, p2, None, None, K)`` for K in {1, 3}, bit for bit, whatever ``seed`` holds.
    ``seed`` [N,P1,K] int32 is updated in place with the new neighbour indices (pass it to the next
    call on the moved clouds); ``None`` starts cold.  Not part of the reference's ``_C`` module: it is
    the stand-alone form of the search the fused relaxation step runs every iteration.
    -> (idx int64 [N,P1,K], dists float32 [N,P1,K], seed)"""
    _lib.require_gpu(p1, p2)
    if p1.dtype != torch.float32 or p2.dtype != torch.float32:
        raise TypeError("knn_points_idx_warm expects float32 point clouds")
    p1, p2 = p1.contiguous(), p2.contiguous()
    N, P1, _ = p1.shape
    P2 = p2.shape[1]
    if seed is None:
        seed = torch.full((N, P1, K), -1, dtype=torch.int32, device=p1.device)
    if seed.dtype != torch.int32 or not seed.is_contiguous() or tuple(seed.shape) != (N, P1, K):
        raise TypeError("seed must be a contiguous int32 tensor [N,P1,K]")
    dists = torch.empty((N, P1, K), dtype=torch.float32, device=p1.device)
    idx = torch.empty((N, P1, K), dtype=torch.int64, device=p1.device)
    L = _lib.lib()
    nbytes = L.reart_knn_points_warm_workspace_bytes(N, P1, P2, K)
    ws = _lib.workspace(nbytes, p1.device)
    rc = L.reart_knn_points_idx_warm(_lib.ptr(p1), _lib.ptr(p2), N, P1, P2, K, _lib.ptr(seed), _lib.ptr(dists),
                                     _lib.ptr(idx), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_knn_points_idx_warm")
    return idx, dists, seed
