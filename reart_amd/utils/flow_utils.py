"""Host-side mirror of the per-iteration part of the reference's ``utils/flow_utils.py``."""
import torch

from .. import _lib


@torch.no_grad()
def blend_anchor_motion(query_loc, reference_loc, reference_flow, knn, return_mask=False):
    """Inverse-distance blending of the k nearest anchors' flow (utils/flow_utils.py:147-170).
    query_loc [m,3], reference_loc [n,3], reference_flow [n,3]; ``knn`` is a ``KNN`` instance
    (only its ``k`` and distance convention are used: search and blend are one fused call)."""
    _lib.require_gpu(query_loc, reference_loc, reference_flow)
    q = query_loc.contiguous().float()
    r = reference_loc.contiguous().float()
    f = reference_flow.contiguous().float()
    nq, nr, k = q.shape[0], r.shape[0], knn.k
    flow = torch.empty((nq, 3), dtype=torch.float32, device=q.device)
    mask = torch.empty((nq,), dtype=torch.bool, device=q.device)
    L = _lib.lib()
    ws = _lib.workspace(L.reart_blend_anchor_motion_workspace_bytes(nq, nr, k), q.device)
    euclid = 0 if getattr(knn, "_squared", False) else 1
    rc = L.reart_blend_anchor_motion(_lib.ptr(q), _lib.ptr(r), _lib.ptr(f), nq, nr, k, euclid, _lib.ptr(flow),
                                     _lib.ptr(mask), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_blend_anchor_motion")
    return (flow, mask) if return_mask else flow


def normalize_pc_list(pc_list, centroid, scale):
    """utils/flow_utils.py:173-175."""
    return (pc_list - centroid) * scale
