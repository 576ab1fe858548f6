"""oracle/torch_step.py -- TEST INFRASTRUCTURE (never imported by the product path).

The reference's loop body (run_robot.py:154-221, Chamfer [+ flow] branch) written with the tensor
expressions the reference itself issues, on PyTorch-CPU: this is the "reference-style PyTorch CPU path"
that bench.py times beside the HIP engine (`cpu_baseline_torch`), and tests/test_oracle_torch_cpu.py
checks it against the C oracle (oracle/step.py) on the same inputs and noise.

The two native packages the reference calls and does not vendor are replaced by what they compute
("parity unpinned"):
  * chamferdist._C.knn_points_idx / knn_points_backward (utils/chamfer.py:174,206): squared distances by
    direct differences -> torch.cdist(compute_mode="donot_use_mm_for_euclid_dist") ** 2 is NOT the same
    rounding, so the neighbour is found on the cdist matrix and the distance is then re-evaluated as
    sum((p1 - p2[idx]) ** 2) in fp32, which is also what makes autograd produce knn_points_backward's
    formula 2 g (p1 - p2[idx]) for both arguments (utils/chamfer.py:195-209).
  * knn_cuda.KNN(k=3, transpose_mode=True) (utils/flow_utils.py:158): Euclidean distances, ascending.
"""
import math

import torch
import torch.nn.functional as F


def rotation_6d_to_matrix(d6):
    """screw_se3/geo_utils.py:632-651"""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def base_forward(cano_pc, W1, b1, W2, proposal_6d, proposal_t, tau, gumbel=None):
    """networks/model.py:39-70 (seg_head = MLPConv1d(3, (128, P)), networks/blocks.py:99-118: conv1x1 + bias +
    ReLU, conv1x1 without bias).  gumbel: injected noise [N,P] (the body of F.gumbel_softmax, hard=True)."""
    N = cano_pc.shape[0]
    T, P = proposal_6d.shape[:2]
    inp = cano_pc.permute(1, 0).unsqueeze(0)                                   # [1,3,N]
    h = F.relu(F.conv1d(inp, W1.unsqueeze(-1), b1))
    seg = F.conv1d(h, W2.unsqueeze(-1)).squeeze(0).permute(1, 0)               # [N,P]
    if gumbel is None:
        weight = F.gumbel_softmax(seg, tau=tau, hard=True)
    else:
        y_soft = ((seg + gumbel) / tau).softmax(dim=-1)
        index = y_soft.max(dim=-1, keepdim=True)[1]
        y_hard = torch.zeros_like(seg).scatter_(-1, index, 1.0)
        weight = y_hard - y_soft.detach() + y_soft
    rotation = rotation_6d_to_matrix(proposal_6d.reshape(-1, 6))               # [(T-1)P,3,3]
    translation = proposal_t.reshape(-1, 3)
    cano = cano_pc.unsqueeze(0).unsqueeze(1).expand(T, P, N, 3).reshape(-1, N, 3)
    pc = torch.bmm(cano, rotation.transpose(1, 2)) + translation[:, None, :]   # [(T-1)P,N,3]
    pc = pc.reshape(T, P, N, 3)
    pc = (weight.permute(1, 0)[None, :, :, None] * pc).sum(dim=1)              # [T-1,N,3]
    return pc, seg.argmax(dim=-1)


def _nn_sqdist(p1, p2):
    """K = 1 of _knn_points (utils/chamfer.py:140-209): index without grad, distance re-evaluated on the pair."""
    with torch.no_grad():
        idx = torch.cdist(p1, p2, compute_mode="donot_use_mm_for_euclid_dist").argmin(dim=-1)
    nn = torch.gather(p2, 1, idx[..., None].expand(-1, -1, 3))
    return ((p1 - nn) ** 2).sum(-1)


def recon_loss(pc_trans_list, pc_list):
    """networks/loss.py:24-29 over ChamferDistance(bidirectional=True) (utils/chamfer.py:78-123)"""
    return (_nn_sqdist(pc_trans_list, pc_list) + _nn_sqdist(pc_list, pc_trans_list)).sum()


def blend_anchor_motion(query_loc, reference_loc, reference_flow, k=3):
    """utils/flow_utils.py:147-170 with return_mask=True"""
    d = torch.cdist(query_loc[None], reference_loc[None], compute_mode="donot_use_mm_for_euclid_dist")[0]
    dists, idx = d.topk(k, dim=-1, largest=False)
    dists = dists.clamp_min(1e-10)
    weight = 1.0 / dists
    weight = weight / weight.sum(dim=-1, keepdim=True)
    blended = (reference_flow[idx] * weight.reshape(-1, k, 1)).sum(dim=1)
    min_d = dists.min(dim=-1)[0]
    flow_d = (reference_flow[idx] ** 2).sum(-1).max(dim=1)[0]
    return blended, torch.logical_or(min_d <= flow_d, min_d <= 0.05)


def flow_loss(gt_flow, pred_flow, mask, robust=False, smooth_weight=1e-2):
    """networks/loss.py:10-21"""
    if not robust:
        f = F.mse_loss(pred_flow, gt_flow, reduction="none").sum(dim=2)
    else:
        f = F.huber_loss(pred_flow, gt_flow, reduction="none").sum(dim=2)
    smooth = (pred_flow ** 2).sum(dim=2)
    return (mask * f + smooth_weight * torch.logical_not(mask) * smooth).sum()


class TorchRelax:
    """One optimisation instance on PyTorch-CPU: parameters, the two-group Adam of run_robot.py:146-148, and
    `step()` = one pass of run_robot.py:154-221."""

    def __init__(self, cano, pc_list, W1, b1, W2, p6d, pt, cano_idx, refs=None, ref_flows=None, lambda_flow=1.0,
                 robust=False, trans_lr=1e-2, seg_lr=1e-3, n_iter=15000, start_tau=5.0, end_tau=1.0):
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).clone()
        self.cano, self.pc_list = t(cano), t(pc_list)
        self.W1, self.b1, self.W2 = (torch.nn.Parameter(t(x)) for x in (W1, b1, W2))
        self.p6d, self.pt = torch.nn.Parameter(t(p6d)), torch.nn.Parameter(t(pt))
        self.opt = torch.optim.Adam([{"params": [self.p6d, self.pt], "lr": trans_lr},
                                     {"params": [self.W1, self.b1, self.W2], "lr": seg_lr}], lr=1e-3, weight_decay=0.0)
        self.cano_idx = cano_idx
        self.refs = None if refs is None else [t(r) for r in refs]
        self.ref_flows = None if ref_flows is None else [t(r) for r in ref_flows]
        self.lambda_flow, self.robust = lambda_flow, robust
        self.n_iter, self.start_tau, self.end_tau, self.it = n_iter, start_tau, end_tau, 0

    def step(self, gumbel=None, tau=None):
        if tau is None:   # utils/model_utils.py:33-37 at cur_iter = i + 1
            tau = self.end_tau + (self.start_tau - self.end_tau) * (math.cos(math.pi * (self.it + 1) / self.n_iter) + 1.0) * 0.5
        g = None if gumbel is None else torch.as_tensor(gumbel, dtype=torch.float32)
        pc_trans, seg_part = base_forward(self.cano, self.W1, self.b1, self.W2, self.p6d, self.pt, tau, g)
        recon = recon_loss(pc_trans, self.pc_list)
        loss, flow = recon, torch.zeros(())
        if self.refs is not None:   # run_robot.py:194-209
            c = self.cano_idx
            with torch.no_grad():
                query = torch.cat((pc_trans[:c], self.cano[None], pc_trans[c:]), dim=0)[:-1]
                bl = [blend_anchor_motion(q, r, f) for q, r, f in zip(query, self.refs, self.ref_flows)]
                gt, mask = torch.stack([b[0] for b in bl]), torch.stack([b[1] for b in bl])
            comp = torch.cat((pc_trans[:c], self.cano[None], pc_trans[c:]), dim=0)
            flow = self.lambda_flow * flow_loss(gt, comp[1:] - comp[:-1], mask, self.robust)
            loss = loss + flow
        self.opt.zero_grad()
        loss.backward()
        grads = {k: getattr(self, k).grad.detach().clone() for k in ("W1", "b1", "W2", "p6d", "pt")}
        self.opt.step()
        self.it += 1
        return dict(recon=float(recon.detach()), flow=float(flow.detach()), total=float(loss.detach()), tau=tau,
                    pc_trans=pc_trans.detach().numpy(), seg_part=seg_part.numpy(), grads=grads)
