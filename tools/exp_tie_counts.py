#!/usr/bin/env python3
"""How many tight pairs does a ROW have after a re-solve (reart_lap_ties' per-row slots: K)?  The recipe's assignment phase on nao,
the per-row counts of every refresh: histogram of the per-problem maximum and of all rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, numpy as np, torch
import bench
from reart_amd.utils import lap
from reart_amd.networks.model import BaseModel
from reart_amd.relax import RelaxEngine
from reart_amd.run_robot import AssignmentPhase

dev = torch.device("cuda:0")
lap.CANONICAL_TIES = True
g, cano, pcs, c, complete, refs, flows, matches, gt_pairs, t_corr = bench.nao_correspondences(dev)
torch.manual_seed(2)
model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
n_iter = int(os.environ.get("ITERS", 3000))
eng = RelaxEngine(cano, pcs, model, c, refs, flows, n_iter=n_iter, seed=2)
i = eng.capture(steps_per_graph=50)
eng.step(n_iter // 3 - i)
ph = AssignmentPhase(eng, cano, pcs, 4, 5, 0.3)
rowmax, allc, flags = [], [], []
orig = lap.TieBreaker.settle
def settle(self, src, tgt, state, skip=()):
    cnt = self.n_edges.cpu().numpy()
    rowmax.append(cnt.max(1)); allc.append(np.bincount(cnt.reshape(-1).clip(0, 40), minlength=41)); flags.append(self.tie_host.numpy().copy())
    return orig(self, src, tgt, state, skip)
lap.TieBreaker.settle = settle
ph.run(n_iter // 3, n_iter)
rm = np.concatenate(rowmax); al = np.sum(allc, 0); fl = np.concatenate(flags)
print("refreshes", len(rowmax), "problems", len(rm))
print("per-problem largest row count: percentiles 50/90/99/max", np.percentile(rm, [50, 90, 99]).tolist(), rm.max())
print("share of problems with a row above 8 / 12 / 16 pairs:", float((rm > 8).mean()), float((rm > 12).mean()), float((rm > 16).mean()))
print("rows by count (0..20):", al[:21].tolist(), " above 20:", int(al[21:].sum()))
print("flags:", np.bincount(fl, minlength=4).tolist())
