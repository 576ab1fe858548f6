#!/usr/bin/env python3
"""Cold solve by the re-solve machinery: zero potentials, no previous assignment (every row free) -> greedy, row reduction on
many compute units, path searches -- against the raced cold auction, on the recipe's first refresh (9 x 1024^2) and on the
kinematic projection's size (19 x 2048^2, synthetic).  Usage: gpurun -- python tools/exp_cold_jv.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.utils import lap

dev = torch.device("cuda:0")


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0) / reps


for T, N, ds, iters in ((10, 4096, 4, 5000), (20, 4096, 2, 2000), (20, 4096, 4, 2000)):
    eng, seq, model = bench.build_instance(dev, T, N, T // 2, 2, n_iter=15000)
    eng.capture(steps_per_graph=50)
    eng.step(iters)
    cano, pcs = eng.caller_clouds()
    B, n = pcs.shape[0], N // ds
    zero = torch.zeros(1, dtype=torch.long, device=dev)
    si = farthest_point_sample(cano[None], n, start=zero, cuda_mode=True)
    ti = farthest_point_sample(pcs, n, start=zero.expand(B), cuda_mode=True)
    eng.peek_forward()
    sp = index_points(eng.pc_trans, si.expand(B, n)).contiguous()
    tp = index_points(pcs, ti).contiguous()
    cost = lap.cdist(sp, tp)
    (ref, fb), t_auc = timed(lambda: lap.linear_sum_assignment_batch(cost, points=(sp, tp), race=True, return_stats=True))

    def cold_jv():
        st = {"prices": torch.zeros((B, n), dtype=torch.float64, device=dev), "cols": torch.full((B, n), -1, dtype=torch.int32, device=dev)}
        return lap.linear_sum_assignment_points(sp, tp, st, return_stats="full")
    (out, fb2, st), t_jv = timed(cold_jv)
    same = all(np.array_equal(a[1], b[1]) for a, b in zip(out, ref))
    print(f"{B} x {n}^2: raced cold auction {t_auc:.1f} ms (fb {fb}) | cold re-solve machinery {t_jv:.1f} ms (fb {fb2}, same {same}) "
          f"left {st[:, 1].mean():.0f} search steps mean {st[:, 2].mean():.0f} max {st[:, 2].max()} reduction steps {(st[:, 3] >> 8).mean():.0f}")
