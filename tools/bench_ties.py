#!/usr/bin/env python3
"""reart_lap_ties alone: microseconds per call on B problems of n columns (targets = sources + noise), with the potentials as
the solver left them and shifted by a constant (a run's potentials drift: the fp32 filter must not care)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd.utils import lap

dev = torch.device("cuda:0")
B, n = int(os.environ.get("B", 9)), int(os.environ.get("N", 2048))
rng = np.random.default_rng(0)
src = rng.uniform(0, 1, (B, n, 3)).astype(np.float32)
tgt = np.stack([(s + rng.normal(0, 0.01, s.shape)).astype(np.float32)[rng.permutation(n)] for s in src])
src, tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
state = {}
lap.linear_sum_assignment_points(src, tgt, state, device_cols=True)
src2 = src + torch.randn_like(src) * 0.002
lap.linear_sum_assignment_points(src2, tgt, state, device_cols=True)
tb = lap.TieBreaker(B, n, dev)
for shift in (0.0, -50.0, 1e4):
    prices = state["prices"] + shift
    for _ in range(3):
        tb.launch(src2, tgt, state["cols"], prices)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        tb.launch(src2, tgt, state["cols"], prices)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} n={n} potentials shifted by {shift:g}: {1e3 * e0.elapsed_time(e1) / 50:.1f} us per call (two launches + fill + flag copy), "
          f"flags {tb.tie_host.tolist()}, tight pairs {tb.n_edges.sum(1).cpu().tolist()}")
