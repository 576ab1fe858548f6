"""Host-side mirror of the reference's ``utils/chamfer.py`` (ChamferDistance, knn_points,
knn_gather) over the HIP K-NN kernels.

Same names, argument meaning, return shapes and error behaviour as the reference module
(``utils/chamfer.py:20-337``); the native calls at ``:174`` and ``:206-208`` go to
``reart_amd.chamferdist_C`` instead of ``chamferdist._C``.
"""
import warnings
from collections import namedtuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import chamferdist_C as _C

_KNN = namedtuple("KNN", "dists idx knn")


class _knn_points(Function):
    """autograd wrapper, cf. utils/chamfer.py:135-209."""

    @staticmethod
    def forward(ctx, p1, p2, lengths1, lengths2, K, version, return_sorted=True):
        idx, dists = _C.knn_points_idx(p1, p2, lengths1, lengths2, K, version)
        # The HIP kernel already returns neighbours ascending by (distance, index) and
        # zero-fills slots beyond lengths2, which is what the reference's post-sort
        # (utils/chamfer.py:177-189) produces.
        ctx.save_for_backward(p1, p2, lengths1, lengths2, idx)
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_dists, grad_idx):
        p1, p2, lengths1, lengths2, idx = ctx.saved_tensors
        grad_p1, grad_p2 = _C.knn_points_backward(
            p1.float(), p2.float(), lengths1, lengths2, idx, grad_dists.float()
        )
        return grad_p1, grad_p2, None, None, None, None, None


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False, return_sorted=True):
    """K nearest neighbours of every p1 point in p2 (utils/chamfer.py:212-286).

    Returns the namedtuple ``(dists [N,P1,K] squared, idx [N,P1,K] int64, knn or None)``.
    """
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    p1, p2 = p1.contiguous(), p2.contiguous()
    n = p1.shape[0]
    if lengths1 is None:
        lengths1 = torch.full((n,), p1.shape[1], dtype=torch.int64, device=p1.device)
    if lengths2 is None:
        lengths2 = torch.full((n,), p2.shape[1], dtype=torch.int64, device=p1.device)
    dists, idx = _knn_points.apply(p1, p2, lengths1, lengths2, K, version, return_sorted)
    nn = knn_gather(p2, idx, lengths2) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)


def knn_gather(x, idx, lengths=None):
    """x [N,M,U], idx [N,L,K] -> [N,L,K,U] with x_out[n,l,k] = x[n, idx[n,l,k]]
    (utils/chamfer.py:289-337); entries with k >= lengths[n] are zero."""
    N, M, U = x.shape
    n2, L, K = idx.shape
    if N != n2:
        raise ValueError("x and idx must have same batch dimension.")
    if lengths is None:
        lengths = torch.full((N,), M, dtype=torch.int64, device=x.device)
    out = x.gather(1, idx.reshape(N, L * K, 1).expand(-1, -1, U)).reshape(N, L, K, U)
    if lengths.min() < K:
        dead = lengths[:, None] <= torch.arange(K, device=x.device)[None]
        out = out.masked_fill(dead[:, None, :, None], 0.0)
    return out


class ChamferDistance(torch.nn.Module):
    """Per-point (un-reduced) Chamfer distance, cf. utils/chamfer.py:19-132.

    ``reduction`` is validated and then ignored, exactly like the reference (its reduction
    block is commented out, utils/chamfer.py:104-117): the result has shape [B, P].
    """

    def forward(self, source_cloud, target_cloud, bidirectional=False, reverse=False,
                reduction="mean", return_index=False):
        for cloud in (source_cloud, target_cloud):
            if not isinstance(cloud, torch.Tensor):
                raise TypeError("Expected input type torch.Tensor. Got {} instead".format(type(cloud)))
        if source_cloud.device != target_cloud.device:
            raise ValueError(
                "Source and target clouds must be on the same device. "
                f"Got {source_cloud.device} and {target_cloud.device}."
            )
        bs, ns, ds = source_cloud.shape
        bt, nt, dt = target_cloud.shape
        if bs != bt:
            raise ValueError("Source and target pointclouds must have the same batchsize.")
        if ds != dt:
            raise ValueError("Source and target pointclouds must have the same dimensionality.")
        if bidirectional and reverse:
            warnings.warn("Both bidirectional and reverse set to True. bidirectional behavior takes precedence.")
        if reduction not in ("sum", "mean"):
            raise ValueError('Reduction must either be "sum" or "mean".')

        len_s = torch.full((bs,), ns, dtype=torch.long, device=source_cloud.device)
        len_t = torch.full((bt,), nt, dtype=torch.long, device=target_cloud.device)
        fwd = knn_points(source_cloud, target_cloud, lengths1=len_s, lengths2=len_t, K=1)
        d_fwd, i_fwd = fwd.dists[..., 0], fwd.idx[..., 0]
        if reverse or bidirectional:
            bwd = knn_points(target_cloud, source_cloud, lengths1=len_t, lengths2=len_s, K=1)
            d_bwd, i_bwd = bwd.dists[..., 0], bwd.idx[..., 0]
        if bidirectional:
            total = d_fwd + d_bwd  # element-wise: needs ns == nt (utils/chamfer.py:119-123)
            return (total, i_fwd, i_bwd) if return_index else total
        if reverse:
            return (d_bwd, i_bwd) if return_index else d_bwd
        return (d_fwd, i_fwd) if return_index else d_fwd
