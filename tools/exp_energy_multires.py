#!/usr/bin/env python3
"""Cold 19 x 4096^2 assignment of the model-selection energy (utils/model_utils.py:92-104): does a coarse solve help?
A 1024-point FPS subset of both sides is solved first (raced auction); its column potentials are lifted to all 4096 columns
(each takes the potential of its nearest sampled target) and start the full auction (reart_lap_auction with price_in).  The
certificate makes any start safe; the question is only the time.  Usage: gpurun -- python tools/exp_energy_multires.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reart_amd import tail
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.utils.lap import cdist, linear_sum_assignment_batch
from reart_amd.utils.model_utils import compute_pc_transform
from reart_amd.chamferdist_C import knn_points_idx

dev = torch.device("cuda:0")
eng, seq, model = bench.build_instance(dev, 20, 4096, 10, 2, n_iter=15000)
eng.capture(steps_per_graph=50)
eng.step(int(os.environ.get("ITERS", 1650)))
cano, pcs = eng.caller_clouds()
with torch.no_grad():
    _, seg0, trans0 = model(cano)
seg_s, trans_s, conn_s = tail.extract_structure(seg0, trans0, cano)
pred = compute_pc_transform(cano, trans_s, seg_s).contiguous()          # [19,4096,3]
B, N = pred.shape[:2]


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0) / reps


cost = cdist(pred, pcs)
(ref, fb0), t_cold = timed(lambda: linear_sum_assignment_batch(cost, points=(pred, pcs), race=True, return_stats=True))
print(f"cold raced auction 19 x 4096^2: {t_cold:.1f} ms, fallbacks {fb0}")
for m in (512, 1024, 2048):
    def multires():
        zero = torch.zeros(B, dtype=torch.long, device=dev)
        si = farthest_point_sample(pred, m, start=zero, cuda_mode=True)
        ti = farthest_point_sample(pcs, m, start=zero, cuda_mode=True)
        sp, tp = index_points(pred, si).contiguous(), index_points(pcs, ti).contiguous()
        st = {}
        linear_sum_assignment_batch(cdist(sp, tp), points=(sp, tp), race=True, state=st)
        nn, _ = knn_points_idx(pcs, tp, None, None, 1)                   # nearest sampled target of every target
        st_full = {"prices": torch.gather(st["prices"], 1, nn[..., 0]).contiguous()}
        return linear_sum_assignment_batch(cost, points=(pred, pcs), state=st_full, return_stats="full")
    (out, fb, stt), t = timed(multires, reps=2)
    same = all(np.array_equal(a[1], b[1]) for a, b in zip(out, ref))
    print(f"coarse {m}: {t:.1f} ms total, fallbacks {fb}, same assignment {same}, phases/rounds/bids/cert mean {stt.mean(0).round(0).tolist()}")
