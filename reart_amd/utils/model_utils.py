"""Host-side mirror of the hot-path helpers of the reference's ``utils/model_utils.py``."""
import math

import numpy as np

import torch

from .. import _lib


def tau_cosine(cur_iter, max_iter, end_temp, start_temp):
    """Cosine temperature schedule (utils/model_utils.py:33-37)."""
    assert end_temp <= start_temp
    return end_temp + (start_temp - end_temp) * (math.cos(math.pi * cur_iter / max_iter) + 1.0) * 0.5


def th_with_zeros(tensor):
    """[B,3,4] -> [B,4,4] by appending the row (0,0,0,1) (utils/model_utils.py:12-19)."""
    pad = tensor.new_tensor([0.0, 0.0, 0.0, 1.0]).view(1, 1, 4).expand(tensor.shape[0], 1, 4)
    return torch.cat([tensor, pad], dim=1)


def create_transformation(rotation, translation):
    """rotation [B,3,3], translation [B,3,1] -> [B,4,4] (utils/model_utils.py:22-30)."""
    top = torch.cat([rotation, translation], dim=2)
    return th_with_zeros(top)


def compute_pc_transform(cano_pc, pose_list, cano_part):
    """Apply each point's part transform (utils/model_utils.py:54-67):
    cano_pc [N,3], pose_list [T-1,P,4,4], cano_part [N] -> [T-1,N,3]."""
    _lib.require_gpu(cano_pc, pose_list, cano_part)
    cano = cano_pc.contiguous().float()
    pose = pose_list.contiguous().float()
    part = cano_part.contiguous().long()
    B, P = pose.shape[:2]
    N = cano.shape[0]
    out = torch.empty((B, N, 3), dtype=torch.float32, device=cano.device)
    rc = _lib.lib().reart_compute_pc_transform(_lib.ptr(cano), _lib.ptr(pose), _lib.ptr(part), N, P, B,
                                               _lib.ptr(out), _lib.stream())
    _lib.check(rc, "reart_compute_pc_transform")
    return out


def _row_mode(values):
    """torch.mode(values, dim=1)[0] with the CPU rule spelled out: the most frequent value of each row,
    ties -> the smallest value (the device implementation of torch.mode does not promise a tie rule).
    values [n, k] integer labels; k is small (1 / 3 / 20)."""
    counts = (values[:, :, None] == values[:, None, :]).sum(dim=2)                      # [n, k]
    best = counts.max(dim=1, keepdim=True).values
    big = torch.iinfo(values.dtype).max
    return torch.where(counts == best, values, torch.full_like(values, big)).min(dim=1).values


def knn_query(query_pc, src_pc, src_input, knn):
    """Label / feature transfer from the k nearest source points (utils/model_utils.py:41-51).
    1-D ``src_input`` (labels): per-query mode over the k neighbours.  2-D ``src_input`` [n_src, C]: mean over
    the k neighbours -- the reference reshapes by ``src_input.shape[0]``, i.e. that branch requires as many
    queries as source points, and raises otherwise; so does this."""
    _lib.require_gpu(query_pc, src_pc, src_input)
    _, idx = knn(ref=src_pc.unsqueeze(0), query=query_pc.unsqueeze(0))  # [1, nq, k]
    idx = idx.squeeze(0).reshape(-1)
    if src_input.dim() == 2:
        return src_input[idx].reshape(src_input.shape[0], knn.k, src_input.shape[1]).mean(dim=1)
    return _row_mode(src_input[idx].reshape(-1, knn.k))


# ---------------------------------------------------------------------------------------------------------
# Energy terms of the model selection (run_robot.py:306-321)
def parallel_lap(cost, nproc=None):
    """utils/model_utils.py:85-89: the reference's process pool of host solvers -> one batched GPU solve
    (``nproc`` is accepted and ignored).  cost [T,n,n] tensor or array -> list of (row_ind, col_ind)."""
    from .lap import linear_sum_assignment_batch

    return linear_sum_assignment_batch(torch.as_tensor(cost))


def compute_ass_err(pc_trans_list, pc_list, use_nproc=True):
    """utils/model_utils.py:92-104: mean squared distance under each frame's optimal one-to-one assignment
    (Euclidean cost)."""
    from .lap import cdist, linear_sum_assignment_batch

    _lib.require_gpu(pc_trans_list, pc_list)
    with torch.no_grad():
        cost = cdist(pc_trans_list, pc_list)
        assign, fallbacks = linear_sum_assignment_batch(cost, points=(pc_trans_list, pc_list), race=True, return_stats=True)
        compute_ass_err.last_fallbacks = int(fallbacks)        # matrices of this call that went to the host solver (0 = none)
        cols = torch.from_numpy(np.stack([c for _, c in assign])).to(pc_list.device)
        matched = torch.gather(pc_list, 1, cols[..., None].expand(-1, -1, 3))
        return ((pc_trans_list - matched) ** 2).sum(dim=-1).mean()


def compute_group_temporal_err(pc_list, seg_part):
    """utils/model_utils.py:107-118: the worst part's mean squared distance to its per-frame centroid."""
    _lib.require_gpu(pc_list, seg_part)
    pcs = pc_list.detach().contiguous().float()
    seg = seg_part.contiguous().long()
    lab = torch.unique(seg, sorted=True)
    per = torch.empty((lab.shape[0],), dtype=torch.float32, device=pcs.device)
    worst = torch.empty((1,), dtype=torch.float32, device=pcs.device)
    rc = _lib.lib().reart_group_temporal_err(_lib.ptr(pcs), pcs.shape[0], pcs.shape[1], _lib.ptr(seg), _lib.ptr(lab),
                                             lab.shape[0], _lib.ptr(per), _lib.ptr(worst), _lib.stream())
    _lib.check(rc, "reart_group_temporal_err")
    return worst[0].cpu()


def compute_align_trans(trans_list, root_trans):
    """utils/model_utils.py:121-126: express every part's motion in the root's frame."""
    from ..screw_se3 import inverse_transformation

    return torch.matmul(inverse_transformation(root_trans)[:, None, :, :], trans_list)
