"""oracle/kinematic_step.py -- TEST INFRASTRUCTURE (never imported by reart_amd).

The reference's kinematic projection iteration (run_robot.py:154-221 with `--model kinematic --use_assign_loss
--assign_iter 0 [--use_flow_loss]`, README.md:125) on the host, from the ORACLE's pieces:

  forward kinematics     torch restatement below of screw_se3/screw_utils.py:6-30, screw_se3/geo_utils.py:90-222 and
                         utils/kinematic_utils.py:151-198 (fp32 CPU tensors, autograd supplies the backward, as in the
                         reference); pinned to tests/golden/kinematic.npz -- the reference's own forward AND autograd
                         gradients on its shipped kinematic-2 checkpoint (tests/test_oracle_golden_cpu.py)
  rigid apply            networks/model.py:144-166 (seg by k-NN label transfer = the stored labels on the stored cloud)
  FPS subsets            oracle.fps, the reference's CUDA rule (start 0), run_robot.py:167-169
  cost matrices          oracle.cdist (run_robot.py:171)
  optimal assignment     scipy.optimize.linear_sum_assignment, the reference's own call (run_robot.py:172-176)
  assignment loss        run_robot.py:181-184
  flow blend + loss      oracle.blend_anchor_motion (C), flow_loss of oracle/torch_step.py (networks/loss.py:10-21)
  Adam                   oracle.adam (numpy restatement of torch.optim.Adam's defaults, run_robot.py:149-151)
"""
import math

import numpy as np
import torch

import oracle
from oracle.torch_step import flow_loss


def _hat(w):
    z = torch.zeros_like(w[..., 0])
    return torch.stack([torch.stack([z, -w[..., 2], w[..., 1]], -1), torch.stack([w[..., 2], z, -w[..., 0]], -1),
                        torch.stack([-w[..., 1], w[..., 0], z], -1)], -2)


def se3_exp(v, w):
    """exp of the twist [v | w] -> (R [...,3,3], t [...,3]); screw_se3/geo_utils.py:90-144 (_so3_exp_map, clamp of the
    SQUARED norm at 1e-4) and :202-222 (_se3_V_matrix)."""
    n2 = (w * w).sum(-1).clamp(min=1e-4)
    ph = n2.sqrt()
    s, c = torch.sin(ph), torch.cos(ph)
    K = _hat(w)
    K2 = K @ K
    eye = torch.eye(3, dtype=w.dtype).expand(K.shape)
    f = lambda x: x[..., None, None]
    R = f(s / ph) * K + f((1.0 - c) / n2) * K2 + eye
    V = eye + K * f((1.0 - c) / n2) + K2 * f((ph - s) / (ph * ph * ph))
    return R, (V @ v[..., None])[..., 0]


def screw_to_transform(l, m, theta, d):
    """screw_se3/screw_utils.py:6-30: (axis l, moment m, angle theta, distance d) -> column-vector 4x4."""
    no_rot = (theta.abs() < 1e-6) | ((theta - math.pi).abs() < 1e-6)            # strict, fp32
    q = torch.cross(l, m, dim=-1)
    safe = torch.where(no_rot, torch.ones_like(theta), theta)
    v_rot = torch.cross(q, l, dim=-1) + (d / safe)[..., None] * l
    w = torch.where(no_rot[..., None], torch.zeros_like(l), l)
    v = torch.where(no_rot[..., None], l, v_rot)
    R, t = se3_exp(v * theta[..., None], w * theta[..., None])
    top = torch.cat([R, t[..., None]], -1)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=R.dtype).expand(top.shape[:-2] + (1, 4))
    return torch.cat([top, bottom], -2)


def fk(parent, edge_of_part, order, axis, moment, theta, distance=None):
    """utils/kinematic_utils.py:151-198 with the joint tree as arrays (oracle.fk's convention): parents precede children
    in `order`, so FK[c] = FK[parent] @ T_rel(edge c); revolute joints carry d = 1e-6 (:176).  -> [B,P,4,4]."""
    B, E = theta.shape
    d = torch.full_like(theta, 1e-6) if distance is None else distance
    T = screw_to_transform(axis[None].expand(B, E, 3), moment[None].expand(B, E, 3), theta, d)     # [B,E,4,4]
    out = [None] * len(parent)
    for c in order:
        c = int(c)
        out[c] = torch.eye(4, dtype=theta.dtype).expand(B, 4, 4) if parent[c] < 0 else out[int(parent[c])] @ T[:, int(edge_of_part[c])]
    return torch.stack(out, 1)


def apply_parts(cano, trans, seg):
    """networks/model.py:160-165: every canonical point moved by the transform of its part -> [B,N,3]."""
    Tn = trans[:, seg]                                                                            # [B,N,4,4]
    return (Tn[..., :3, :3] @ cano[None, :, :, None])[..., 0] + Tn[..., :3, 3]


class KinematicOracle:
    """One kinematic projection instance on the host; `iteration(i)` = one pass of run_robot.py:154-221."""

    def __init__(self, cano, pc_list, seg, parent, edge_of_part, order, axis, moment, theta, cano_idx, refs=None, ref_flows=None,
                 trans_lr=1e-2, assign_gap=1, downsample=2, lambda_assign=0.3, lambda_flow=1.0, robust=False, nproc=1):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).clone()
        self.cano, self.pc_list, self.seg = f32(cano), f32(pc_list), torch.as_tensor(np.asarray(seg)).long()
        self.parent, self.eop, self.order = (np.asarray(x) for x in (parent, edge_of_part, order))
        self.params = [f32(axis).requires_grad_(), f32(moment).requires_grad_(), f32(theta).requires_grad_()]
        self.m = [np.zeros(p.shape, np.float32) for p in self.params]
        self.v = [np.zeros(p.shape, np.float32) for p in self.params]
        self.steps, self.lr = 0, trans_lr
        self.cano_idx, self.gap, self.lam_a, self.lam_f, self.robust = cano_idx, assign_gap, lambda_assign, lambda_flow, robust
        self.refs = None if refs is None else [np.asarray(r, np.float32) for r in refs]
        self.ref_flows = None if ref_flows is None else [np.asarray(r, np.float32) for r in ref_flows]
        B, N = self.pc_list.shape[:2]
        n = N // downsample
        # run_robot.py:167-169 on the reference's CUDA path: FPS starts at index 0 and samples fixed clouds
        self.src_idx = torch.from_numpy(oracle.fps(self.cano.numpy()[None], n, start=np.zeros(1, np.int64), cuda_mode=True)[0]).long()
        tgt_idx = oracle.fps(self.pc_list.numpy(), n, start=np.zeros(B, np.int64), cuda_mode=True)
        self.tgt_pts = torch.from_numpy(np.take_along_axis(self.pc_list.numpy(), tgt_idx[..., None], axis=1))
        self.matched = self.cols = None
        self.nproc = nproc          # > 1: the reference's --use_nproc pool (utils/model_utils.py:85-89)
        self.lap_solves = 0

    def forward(self):
        trans = fk(self.parent, self.eop, self.order, *self.params)
        return apply_parts(self.cano, trans, self.seg), trans

    def iteration(self, i):
        for p in self.params:
            p.grad = None
        pc_trans, _ = self.forward()
        pc_src = pc_trans[:, self.src_idx]
        if self.matched is None or i % self.gap == 0:
            cost = oracle.cdist(pc_src.detach().numpy(), self.tgt_pts.numpy())
            solve = oracle.linear_sum_assignment if self.nproc <= 1 else (lambda c: oracle.parallel_lap(c, self.nproc))
            self.cols = np.stack([c for _, c in solve(cost)])
            self.matched = torch.from_numpy(np.take_along_axis(self.tgt_pts.numpy(), self.cols[..., None], axis=1))
            self.lap_solves += 1
        ass = self.lam_a * ((pc_src - self.matched) ** 2).sum(-1).sum()
        losses, loss = {"opt assignment loss": ass}, ass
        if self.refs is not None:
            c = self.cano_idx
            comp = torch.cat((pc_trans[:c], self.cano[None], pc_trans[c:]), dim=0)
            q = comp.detach().numpy()
            bl = [oracle.blend_anchor_motion(q[f], r, fl, k=3) for f, (r, fl) in enumerate(zip(self.refs, self.ref_flows))]
            gt = torch.from_numpy(np.stack([b[0] for b in bl]))
            mask = torch.from_numpy(np.stack([b[1] for b in bl]).astype(bool))
            fl = self.lam_f * flow_loss(gt, comp[1:] - comp[:-1], mask, self.robust)
            losses["flow Loss"] = fl
            loss = loss + fl
        losses["total Loss"] = loss
        loss.backward()
        self.steps += 1
        self.grads = [p.grad.detach().numpy().copy() for p in self.params]
        for p, g, m, v in zip(self.params, self.grads, self.m, self.v):
            oracle.adam(p.detach().numpy(), g, m, v, self.steps, self.lr)        # in place (the array shares the tensor's memory)
        return {k: float(v.detach()) for k, v in losses.items()}, pc_trans.detach().numpy()
