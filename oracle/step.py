"""oracle/step.py -- TEST INFRASTRUCTURE.  One relaxation iteration of the reference's loop
(run_robot.py:154-221, Chamfer [+ flow] branch) assembled from the oracle's C functions; used
as the checker of the fused HIP step and as bench.py's cpu_baseline ("port")."""
import math

import numpy as np

from . import (adam, base_backward, base_forward, blend_anchor_motion, flow_loss, knn_points,
               knn_points_backward)


def tau_cosine(cur_iter, max_iter, end_temp, start_temp):
    """utils/model_utils.py:33-37"""
    return end_temp + (start_temp - end_temp) * (math.cos(math.pi * cur_iter / max_iter) + 1.0) * 0.5


class RelaxOracle:
    def __init__(self, cano, pc_list, W1, b1, W2, p6d, pt, cano_idx, refs=None, ref_flows=None,
                 lambda_flow=1.0, robust=False, smooth_weight=1e-2, trans_lr=1e-2, seg_lr=1e-3,
                 n_iter=15000, start_tau=5.0, end_tau=1.0, euclidean=True, weight_decay=0.0):
        f = lambda a: np.array(a, dtype=np.float32, copy=True, order="C")
        self.cano, self.pc_list = f(cano), f(pc_list)
        self.params = dict(W1=f(W1), b1=f(b1), W2=f(W2), p6d=f(p6d), pt=f(pt))
        self.m = {k: np.zeros_like(v) for k, v in self.params.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.params.items()}
        self.cano_idx, self.refs, self.ref_flows = cano_idx, refs, ref_flows
        self.lambda_flow, self.robust, self.smooth = lambda_flow, robust, smooth_weight
        self.lr = dict(W1=seg_lr, b1=seg_lr, W2=seg_lr, p6d=trans_lr, pt=trans_lr)
        self.n_iter, self.start_tau, self.end_tau, self.euclidean = n_iter, start_tau, end_tau, euclidean
        self.weight_decay = np.float32(weight_decay)     # torch.optim.Adam(weight_decay=...): grad += wd * param (run_robot.py:146-148)
        self.it = 0

    def step(self, gumbel, tau=None, assign=None):
        """assign = (src_idx [n], tgt_idx [B,n], lambda_assign): the assignment loss of run_robot.py:181-184
        replaces the Chamfer loss (pairs fixed by the caller)."""
        p = self.params
        if tau is None:
            tau = tau_cosine(self.it + 1, self.n_iter, self.end_tau, self.start_tau)
        tau = float(np.float32(tau))
        fw = base_forward(self.cano, p["W1"], p["b1"], p["W2"], p["p6d"], p["pt"], gumbel, tau)
        X, Y = fw["out"], self.pc_list
        B, N = X.shape[:2]
        if assign is not None:
            src, tgt, lam = assign
            lam = np.float32(lam)
            G = np.zeros_like(X)
            recon = 0.0
            for b_ in range(B):
                d = X[b_, src] - Y[b_, tgt[b_]]
                recon += float(((d * d)[:, 0] + (d * d)[:, 1] + (d * d)[:, 2]).astype(np.float64).sum())
                G[b_, src] = lam * (np.float32(2.0) * d)
            recon *= float(lam)
        else:
            # recon_loss (networks/loss.py:24-29) and its gradient w.r.t. pc_trans
            d1, i1 = knn_points(X, Y)
            d2, i2 = knn_points(Y, X)
            recon = float((d1[..., 0] + d2[..., 0]).astype(np.float64).sum())
            ones = np.ones((B, N, 1), np.float32)
            gx1, _ = knn_points_backward(X, Y, i1, ones)
            _, gx2 = knn_points_backward(Y, X, i2, ones)
            G = gx1 + gx2
        flow = 0.0
        if self.refs is not None:
            # run_robot.py:194-209
            comp = np.concatenate([X[: self.cano_idx], self.cano[None], X[self.cano_idx:]], axis=0)
            gtf = np.empty((B, N, 3), np.float32)
            msk = np.empty((B, N), bool)
            for f_ in range(B):
                gtf[f_], msk[f_] = blend_anchor_motion(comp[f_], self.refs[f_], self.ref_flows[f_], 3, self.euclidean)
            pred = comp[1:] - comp[:-1]
            fl, gpf = flow_loss(gtf, pred, msk, self.robust, self.smooth)
            flow = self.lambda_flow * fl
            gpf = gpf * np.float32(self.lambda_flow)
            gcomp = np.zeros_like(comp)
            gcomp[1:] += gpf
            gcomp[:-1] -= gpf
            G = G + np.concatenate([gcomp[: self.cano_idx], gcomp[self.cano_idx + 1:]], axis=0)
        g = base_backward(self.cano, p["W1"], p["b1"], p["W2"], p["p6d"], p["pt"], fw["y_soft"], fw["hard_idx"], tau, G)
        self.it += 1
        for k, gk in (("W1", "gW1"), ("b1", "gb1"), ("W2", "gW2"), ("p6d", "g6d"), ("pt", "gt")):
            gr = g[gk].reshape(-1)
            if self.weight_decay != 0:
                gr = (gr + self.weight_decay * p[k].reshape(-1)).astype(np.float32)
            adam(p[k].reshape(-1), gr, self.m[k].reshape(-1), self.v[k].reshape(-1), self.it, self.lr[k])
        return dict(recon=recon, flow=flow, total=recon + flow, tau=tau, pc_trans=X, G=G, grads=g,
                    seg_part=fw["seg_part"], hard_idx=fw["hard_idx"])
