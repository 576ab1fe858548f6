#!/usr/bin/env python3
"""One refresh cycle of the assignment-loss phase (run_robot.py:164-187 as reart_amd/run_robot.py runs it), stage by
stage, at T = 20 x N = 4096, downsample 4 (19 matrices of 1024 x 1024), assign_gap 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.utils.lap import cdist, linear_sum_assignment_batch

dev = torch.device("cuda:0")
eng, seq, model = bench.build_instance(dev, 20, 4096, 10, 2, n_iter=15000)
eng.capture(steps_per_graph=10)
eng.step(5000); torch.cuda.synchronize()
cano, pcs = eng.cano, eng.pc_list
B, N = pcs.shape[:2]; nf = N // 4
acc = {}
def T(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r
WARM = os.environ.get('WARM', '0')      # 4: race with warm racers, 0: cold solves, 1: warm potentials, 2: re-solve from the previous optimum (paths), 3: points form of 2
state = {} if WARM != '0' else None
reps = int(os.environ.get('REPS', 30))
t_all = time.perf_counter()
for k in range(reps):
    T("peek_forward", eng.peek_forward)
    pred = eng.pc_trans
    src = T("fps cano", lambda: farthest_point_sample(cano[None], nf))
    tgt = T("fps frames", lambda: farthest_point_sample(pcs, nf))
    cost = T("gather + cdist (reart_cdist)", lambda: cdist(index_points(pred, src.expand(B, nf)), index_points(pcs, tgt)))
    T("  (torch.cdist, for comparison)", lambda: torch.cdist(index_points(pred, src.expand(B, nf)), index_points(pcs, tgt)))
    if WARM == '5':      # Jonker-Volgenant from scratch in the points form: no previous assignment, zero potentials
        from reart_amd.utils.lap import linear_sum_assignment_points
        pa, pb = index_points(pred, src.expand(B, nf)).contiguous(), index_points(pcs, tgt).contiguous()
        jv = {"prices": torch.zeros((B, nf), dtype=torch.float64, device=dev), "cols": torch.full((B, nf), -1, dtype=torch.int32, device=dev)}
        assign, fb, st = T("linear_sum_assignment_batch", lambda: linear_sum_assignment_points(pa, pb, jv, return_stats="full"))
    elif WARM == '3':
        from reart_amd.utils.lap import linear_sum_assignment_points
        pa, pb = index_points(pred, src.expand(B, nf)).contiguous(), index_points(pcs, tgt).contiguous()
        assign, fb, st = T("linear_sum_assignment_batch", lambda: linear_sum_assignment_points(pa, pb, state, return_stats="full"))
    else:
        assign, fb, st = T("linear_sum_assignment_batch", lambda: linear_sum_assignment_batch(cost, return_stats="full", state=state, warm_assignment=WARM == '2', race=('warm' if WARM == '4' else os.environ.get('RACE', '1') == '1')))
    if k % 5 == 0:
        print(f"cycle {k}: LAP mean rounds {st[:,1].mean():.0f} bids {st[:,2].mean():.0f} (max {st[:,2].max()}) cert {st[:,3].mean():.0f} fallbacks {fb}")
    cols = T("cols to device", lambda: torch.from_numpy(np.stack([c for _, c in assign])).to(dev))
    T("set_assignment", lambda: eng.set_assignment(src[0], tgt.gather(1, cols), 0.3))
    T("5 iterations", lambda: eng.step(5))
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
for k, v in acc.items():
    print(f"{k:32s} {v / reps * 1e3:8.2f} ms")
print(f"{'cycle (with the syncs above)':32s} {tot / reps * 1e3:8.2f} ms")
