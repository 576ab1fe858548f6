"""Drop-in for the ``knn_cuda`` package (``from knn_cuda import KNN``, reference
``run_robot.py:14``): ``KNN(k, transpose_mode)(ref, query) -> (dist, idx)``.

Call sites and shape contract: ``run_robot.py:65-66,122,138``, ``utils/model_utils.py:42``
(``transpose_mode=True``: ref [b,nr,dim], query [b,nq,dim] -> [b,nq,k]) and
``utils/flow_utils.py:127`` (``transpose_mode=False``: [b,dim,n] -> [b,k,nq]).
Distances are Euclidean (sqrt of the squared distance) like upstream KNN_CUDA 0.2; pass
``squared=True`` to get squared distances (the reference is silent on this).
No gradient flows through it.
"""
import torch

from . import _lib


class KNN(torch.nn.Module):
    def __init__(self, k, transpose_mode=False, squared=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode
        self._squared = squared

    @torch.no_grad()
    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        _lib.require_gpu(ref, query)
        if not self._t:  # [b, dim, n] -> [b, n, dim]
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        ref, query = ref.contiguous().float(), query.contiguous().float()
        B, nr, D = ref.shape
        nq = query.shape[1]
        dist = torch.empty((B, nq, self.k), dtype=torch.float32, device=ref.device)
        idx = torch.empty((B, nq, self.k), dtype=torch.int64, device=ref.device)
        L = _lib.lib()
        ws = _lib.workspace(L.reart_knn_points_workspace_bytes(B, nq, nr, self.k), ref.device)
        rc = L.reart_knn_cuda(_lib.ptr(ref), _lib.ptr(query), B, nr, nq, D, self.k, 0 if self._squared else 1,
                              _lib.ptr(dist), _lib.ptr(idx), _lib.ptr(ws), ws.numel(), _lib.stream())
        _lib.check(rc, "reart_knn_cuda")
        if not self._t:  # [b, nq, k] -> [b, k, nq]
            dist, idx = dist.transpose(1, 2).contiguous(), idx.transpose(1, 2).contiguous()
        return dist, idx
