"""Host-side mirror of the reference's ``networks/loss.py`` (recon_loss ``:24-29``, flow_loss
``:10-21``) over the HIP kernels."""
import torch

from .. import _lib


def recon_loss(pc_trans_list, pc_list, chamfer_dist):
    """Sum of the bidirectional per-point Chamfer distance (networks/loss.py:24-29).
    pc_trans_list, pc_list: [T-1, N, 3]."""
    cd = chamfer_dist(pc_trans_list, pc_list, bidirectional=True)  # [T-1, N]
    return torch.sum(cd)


class _FlowLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, pred, mask, robust, smooth_weight):
        _lib.require_gpu(gt, pred, mask)
        gt, pred = gt.contiguous().float(), pred.contiguous().float()
        B, N, _ = pred.shape
        m = None if mask is None else (mask != 0).contiguous()  # bool: one byte per point
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred)
        L = _lib.lib()
        ws = _lib.workspace(L.reart_flow_loss_workspace_bytes(), pred.device)
        rc = L.reart_flow_loss(_lib.ptr(gt), _lib.ptr(pred), _lib.ptr(m), B, N, int(bool(robust)),
                               float(smooth_weight), _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws), ws.numel(),
                               _lib.stream())
        _lib.check(rc, "reart_flow_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        gp = grad * g
        return None, gp, None, None, None


def flow_loss(gt_flow_list, pred_flow_list, flow_mask_list=None, robust=False, smooth_weight=1e-2):
    """networks/loss.py:10-21: masked MSE (or Huber, delta=1) between predicted and blended flow
    plus ``smooth_weight`` x |pred|^2 on the un-masked points; returns the scalar sum.
    Gradient flows to ``pred_flow_list`` (the reference computes ``gt_flow_list`` under no_grad)."""
    if gt_flow_list.requires_grad:
        raise NotImplementedError("flow_loss: gt_flow_list is a constant in the reference (run_robot.py:195)")
    return _FlowLoss.apply(gt_flow_list, pred_flow_list, flow_mask_list, robust, smooth_weight)
