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


@torch.no_grad()
def find_mutual_correspondences(nns01, nns10):
    """utils/flow_utils.py:102-113: the pairs (i, nns01[i]) whose target points back at i."""
    idx0 = torch.arange(len(nns01), device=nns10.device)
    keep = nns10[nns01] == idx0
    return idx0[keep], nns01[keep]


def normalize_pc_list(pc_list, centroid, scale):
    """utils/flow_utils.py:173-175."""
    return (pc_list - centroid) * scale


@torch.no_grad()
def match_smnn(desc1, desc2, th=0.9):
    """Mutual second-nearest-neighbour ratio matching (utils/flow_utils.py:48-100).
    desc1 [B1,D], desc2 [B2,D] -> (match_dists [B3,1], matches_idxs [B3,2]) sorted by the desc1 index.
    (match_dists is the larger of the two ratios like the reference; its callers ignore it.)"""
    if desc1.shape[0] < 2 or desc2.shape[0] < 2:
        raise AssertionError
    keep, tgt = _smnn_batched(desc1[None], desc2[None], th)
    src = torch.nonzero(keep[0], as_tuple=False)[:, 0]
    idx = torch.stack([src, tgt[0][src]], dim=1)
    return torch.empty((idx.shape[0], 1), device=desc1.device), idx


def _smnn_batched(d1, d2, th):
    _lib.require_gpu(d1, d2)
    d1, d2 = d1.contiguous().float(), d2.contiguous().float()
    E, N1, D = d1.shape
    N2 = d2.shape[1]
    keep = torch.empty((E, N1), dtype=torch.bool, device=d1.device)
    tgt = torch.empty((E, N1), dtype=torch.int64, device=d1.device)
    L = _lib.lib()
    ws = _lib.workspace(L.reart_match_smnn_workspace_bytes(E, N1, N2), d1.device)
    rc = L.reart_match_smnn(_lib.ptr(d1), _lib.ptr(d2), E, N1, N2, D, float(th), _lib.ptr(keep), _lib.ptr(tgt),
                            _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_match_smnn")
    return keep, tgt


@torch.no_grad()
def compute_corr_list_filter(norm_pc_list, feature_extractor, knn=None, matching="smnn"):
    """One-time correspondence extraction (utils/flow_utils.py:116-143): descriptors of frames
    [:-1] and [1:], then per pair the mutual SMNN matches.  norm_pc_list [T,N,3] ->
    (corrs_src_list, corrs_tgt_list): ragged index lists per consecutive frame pair.
    Every frame's descriptors are computed once (the reference evaluates interior frames twice)."""
    if matching not in ("smnn", "mnn"):
        raise ValueError("matching is 'smnn' (the reference's default, run_robot.py:80) or 'mnn'")
    feats = feature_extractor(norm_pc_list.transpose(1, 2).contiguous())   # [T,64,N]
    feats = feats.permute(0, 2, 1).contiguous()                            # point-major rows
    # "mnn" (utils/flow_utils.py:126-137): nearest descriptor in both directions + the mutual filter, no ratio test --
    # the same two top-2 passes with the test switched off (th < 0); ``knn`` is not needed (the reference's k = 1 KNN)
    keep, tgt = _smnn_batched(feats[:-1], feats[1:], 0.9 if matching == "smnn" else -1.0)
    src_list, tgt_list = [], []
    for e in range(keep.shape[0]):
        src = torch.nonzero(keep[e], as_tuple=False)[:, 0]
        src_list.append(src)
        tgt_list.append(tgt[e][src])
    return src_list, tgt_list
