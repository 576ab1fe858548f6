#!/usr/bin/env python3
"""Cold solve by the RE-SOLVE machinery (all rows free, zero potentials: team chains + trees + bucketed forest + bucket-round
searches) against the raced epsilon-scaling auction the loops use for their first refresh.  nao's first refresh at both recipe
sizes (9 x 1024^2, 9 x 2048^2).  Round 4 measured the same idea with one-column searches and lone-wave chains: it lost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd.data import load_nao_demo
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.utils import lap

dev = torch.device("cuda:0")
g = load_nao_demo()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cano, pcs = t(g["cano"]), t(g["pc_list"])
B, N = pcs.shape[:2]
for ds in (4, 2):
    n = N // ds
    zero = torch.zeros(1, dtype=torch.long, device=dev)
    src_idx = farthest_point_sample(cano[None], n, start=zero, cuda_mode=True)
    tgt_idx = farthest_point_sample(pcs, n, start=zero.expand(B), cuda_mode=True)
    src = index_points(cano[None].expand(B, N, 3).contiguous(), src_idx.expand(B, n)).contiguous()
    tgt = index_points(pcs, tgt_idx).contiguous()
    for rep in range(2):
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a, fb = lap.linear_sum_assignment_points(src, tgt, st, return_stats=True)
        torch.cuda.synchronize(); t_cold = 1e3 * (time.perf_counter() - t0)
        st2 = {"cols": torch.full((B, n), -1, dtype=torch.int32, device=dev), "prices": torch.zeros((B, n), dtype=torch.float64, device=dev)}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b_, fb2, stats = lap.linear_sum_assignment_points(src, tgt, st2, return_stats="full")
        torch.cuda.synchronize(); t_re = 1e3 * (time.perf_counter() - t0)
        same = all(np.array_equal(x[1], y[1]) for x, y in zip(a, b_))
        print(f"n = {n}: raced auction {t_cold:.1f} ms (fallbacks {fb}) | re-solve from nothing {t_re:.1f} ms (fallbacks {fb2}, rows left "
              f"{stats[:, 1].tolist()}, search steps {stats[:, 2].tolist()}, reduction steps {(stats[:, 3] >> 8).tolist()}) | same assignment {same}")
