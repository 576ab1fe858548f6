"""Host-side mirror of the reference's ``utils/eval_utils.py`` snapshot metrics, on device tensors."""
import torch

from .chamfer import ChamferDistance


def eval_flow(pred_flow_list, gt_flow_list, acc1_thre=0.05, acc2_thre=0.1, as_tensors=False):
    """utils/eval_utils.py:6-22: end-point error, two accuracy levels and the angular error of [T-1,N,3] flows.
    ``as_tensors``: the four values as 0-d device tensors (no host sync) instead of floats."""
    pred, gt = torch.as_tensor(pred_flow_list), torch.as_tensor(gt_flow_list).to(pred_flow_list.device)
    error = torch.sqrt(((pred - gt) ** 2).sum(2) + 1e-20)
    gt_len = torch.sqrt((gt * gt).sum(2) + 1e-20)
    rel = error / gt_len
    acc1 = ((error <= acc1_thre) | (rel <= acc1_thre)).float().mean(1).mean()
    acc2 = ((error <= acc2_thre) | (rel <= acc2_thre)).float().mean(1).mean()
    unit_gt = gt / gt.norm(dim=-1, keepdim=True)
    unit_pred = pred / pred.norm(dim=-1, keepdim=True)
    eps = 1e-7
    dot = (unit_gt * unit_pred).sum(2).clamp(-1 + eps, 1 - eps)
    dot = torch.where(torch.isnan(dot), torch.ones_like(dot), dot)
    angle = torch.acos(dot).mean(1).mean()
    if as_tensors:
        return error.mean(), acc1, acc2, angle
    return float(error.mean()), float(acc1), float(acc2), float(angle)


def eval_seg(gt_segm, pd_segm, as_tensor=False, num_labels=128):
    """utils/eval_utils.py:25-36: Rand index.  The reference compares two N x N co-membership matrices; the same
    count follows from the contingency table: agreeing ordered pairs = N^2 - sum_a n_a^2 - sum_b n_b^2 + 2 sum_ab n_ab^2."""
    gt, pd = gt_segm.long().reshape(-1), pd_segm.long().reshape(-1).to(gt_segm.device)
    n = gt.numel()
    if as_tensor:      # no host sync: a fixed table of ``num_labels`` x ``num_labels`` cells (the caller sizes it from the data it
        # holds: tail.snapshot_values).  A label outside the table would be dropped from the table but not from n^2 -- a wrong
        # index, silently: the result is NaN instead
        s = int(num_labels)
        ar = torch.arange(s, device=gt.device)
        table = (gt[:, None] == ar[None, :]).double().T @ (pd[:, None] == ar[None, :]).double()     # contingency table (no atomics)
        agree = n * n - (table.sum(1) ** 2).sum() - (table.sum(0) ** 2).sum() + 2 * (table ** 2).sum()
        outside = ((gt < 0) | (gt >= s) | (pd < 0) | (pd >= s)).any()
        return torch.where(outside, torch.full_like(agree, float("nan")), agree / (n * n)).float()
    s = int(max(gt.max(), pd.max())) + 1
    table = torch.bincount(gt * s + pd, minlength=s * s).reshape(s, s).double()
    agree = n * n - (table.sum(1) ** 2).sum() - (table.sum(0) ** 2).sum() + 2 * (table ** 2).sum()
    return (agree / (n * n)).float().cpu().numpy()


def compute_chamfer_list(points_set1, points_set2, reduction="sum"):
    """utils/eval_utils.py:39-66 (KD-tree Chamfer per frame) on the GPU search: per frame the sum / mean of the squared
    nearest-neighbour distances of both directions; "mean" / "sum" reduce over frames like the reference."""
    a, b = torch.as_tensor(points_set1).float(), torch.as_tensor(points_set2).float()
    cd = ChamferDistance()
    d12 = cd(a, b)          # [T,N] squared NN distances a -> b
    d21 = cd(b, a)
    if reduction == "mean":
        return float((d12.mean(1) + d21.mean(1)).mean())
    per = d12.sum(1) + d21.sum(1)
    return float(per.sum()) if reduction == "sum" else per.cpu().numpy()
