#!/usr/bin/env python3
"""Experiment (round 2): how much of the pruned search's work comes from neighbouring canonical points that sampled
DIFFERENT parts (Gumbel) and therefore land far apart after the rigid transforms -- fat target boxes of pc_trans
(direction y -> x) and scattered query waves (x -> y, flow)?  Compares the storage order of the moving cloud
(a) canonical k-d order (today) with (b) the same order stably partitioned by the sampled part.
Everything is computed with torch on the GPU from the engine's state; prints per-state statistics."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_instance  # noqa: E402


def boxes_of(P, n=16):          # P [N,3] -> lo, hi [N/n,3]
    Q = P.reshape(-1, n, 3)
    return Q.min(1).values, Q.max(1).values


def lb_point_box(q, lo, hi):    # q [N,3] -> [N,nbox]
    e = torch.maximum(torch.maximum(lo[None] - q[:, None], q[:, None] - hi[None]), torch.zeros((), device=q.device))
    return (e * e).sum(-1)


def wave_stats(q, thr, tgt, tag):
    """q [N,3] queries in wave order, thr [N], tgt [M,3] targets in box order."""
    lo, hi = boxes_of(tgt)
    need = lb_point_box(q, lo, hi) <= thr[:, None]                      # [N, nbox]
    W = need.reshape(-1, 64, need.shape[1])
    cnt = W.sum(1)                                                       # needers per (wave, box)
    union = (cnt > 0).sum(1).float()
    perq = need.sum(1).float()
    dense = (cnt > 16).sum(1).float()
    sp16 = ((cnt > 8) & (cnt <= 16)).sum(1).float()
    sp8 = ((cnt > 0) & (cnt <= 8)).sum(1).float()
    # coarse filter: 4 groups of 16 lanes, box-to-box bound with the group's largest thr
    G = q.reshape(-1, 4, 16, 3)
    glo, ghi = G.min(2).values, G.max(2).values                          # [waves,4,3]
    gthr = thr.reshape(-1, 4, 16).max(2).values
    e = torch.maximum(torch.maximum(lo[None, None] - ghi[:, :, None], glo[:, :, None] - hi[None, None]),
                      torch.zeros((), device=q.device))
    coarse = (((e * e).sum(-1) <= gthr[:, :, None]).any(1)).sum(1).float()
    instr = 100 * dense + 45 * sp16 + 35 * sp8 + 8 * coarse
    print(f"   {tag:28s} union {union.mean():6.1f}  per-query {perq.mean():5.1f}  coarse {coarse.mean():6.1f}  "
          f"dense {dense.mean():5.1f} sp16 {sp16.mean():5.1f} sp8 {sp8.mean():5.1f}  ~instr/wave {instr.mean():7.0f}")
    return float(instr.mean())


def nn(q, t):
    d = torch.cdist(q[None].double(), t[None].double())[0]
    return d.min(1)


def main():
    dev = torch.device("cuda:0")
    eng, seq, model = build_instance(dev, 20, 4096, 10, seed=2)
    done = 0
    for target in (300, 1500, 6000, 14000):
        eng.step(target - 1 - done)
        torch.cuda.synchronize()
        Xold = eng._pc_trans.clone()
        eng.step(1)
        done = target
        torch.cuda.synchronize()
        X, Y, cano = eng._pc_trans, eng.pc_list, eng.cano
        B, N = X.shape[:2]
        # sampled part of every point: which part transform reproduces pc_trans (frame 0)
        Tl = eng.trans_list                                            # [B,P,4,4]
        cand = torch.einsum("pij,nj->pni", Tl[0, :, :3, :3], cano) + Tl[0, :, None, :3, 3]
        part = ((cand - X[0][None]) ** 2).sum(-1).argmin(0)
        order = torch.sort(part, stable=True).indices                  # k-d order inside every part
        runs = int((part[1:] != part[:-1]).sum()) + 1
        print(f"iteration {target}: {len(torch.unique(part))} parts sampled, {runs} runs of equal part along the k-d order "
              f"(mean run {N / runs:.1f} points)")
        tot = {"a": 0.0, "b": 0.0}
        for b in (0, 5, 9, 14, 18):
            # warm start: neighbour indices of the previous iteration give the bound at the new positions
            i_xy = nn(Xold[b], Y[b]).indices
            i_yx = nn(Y[b], Xold[b]).indices
            thr_xy = ((X[b] - Y[b][i_xy]) ** 2).sum(-1)
            thr_yx = ((Y[b] - X[b][i_yx]) ** 2).sum(-1)
            print(f"  frame {b}")
            tot["a"] += wave_stats(X[b], thr_xy, Y[b], "x->y  (a) k-d order")
            tot["b"] += wave_stats(X[b][order], thr_xy[order], Y[b], "x->y  (b) by part")
            tot["a"] += wave_stats(Y[b], thr_yx, X[b], "y->x  (a) k-d order")
            tot["b"] += wave_stats(Y[b], thr_yx, X[b][order], "y->x  (b) by part")
        print(f"  estimated search instructions, (b)/(a): {tot['b'] / tot['a']:.3f}")


if __name__ == "__main__":
    main()
