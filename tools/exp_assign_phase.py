#!/usr/bin/env python3
"""How the assignment problems of the base recipe's second phase (run_robot.py:164-187, README.md:116 on the nao demo:
9 x 1024^2 every 5 iterations, 2 000 refreshes) change from refresh to refresh, and what each solver form costs on them:
every EVERY-th refresh is solved three ways from the same inputs -- raced cold auction, race with warm racers (the
production call), points-form re-solve from the previous refresh's optimum -- and the rows that moved / changed their
column since the previous refresh are counted.  Usage: gpurun -- python tools/exp_assign_phase.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd.networks.model import BaseModel
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.relax import RelaxEngine
from reart_amd.utils.lap import cdist, linear_sum_assignment_batch, linear_sum_assignment_points

dev = torch.device("cuda:0")
from reart_amd.data import load_nao_demo
g = load_nao_demo()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cano, pcs, cano_idx = t(g["cano"]), t(g["pc_list"]), int(g["cano_idx"])
n_iter, assign_iter = int(os.environ.get("ITERS", 15000)), int(os.environ.get("ASSIGN_ITER", 5000))
EVERY = int(os.environ.get("EVERY", 25))
gt_pos = t(g["complete_gt_pc_list"])
rng = np.random.default_rng(0)
sel = [torch.from_numpy(rng.permutation(gt_pos.shape[1])[:3000]).to(dev) for _ in range(pcs.shape[0])]
refs = [gt_pos[k][s] for k, s in enumerate(sel)]
flows = [t(g["gt_flow_list"][k])[s] for k, s in enumerate(sel)]
torch.manual_seed(2)
model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
eng = RelaxEngine(cano, pcs, model, cano_idx, refs, flows, n_iter=n_iter, seed=2)
i = eng.capture(steps_per_graph=10)
eng.step(assign_iter - i); i = assign_iter
B, N = pcs.shape[:2]; nf = N // 4
zero = torch.zeros(1, dtype=torch.long, device=dev)
src = farthest_point_sample(cano[None], nf, start=zero, cuda_mode=True)
tgt = farthest_point_sample(pcs, nf, start=zero.expand(B), cuda_mode=True)
tgt_pts = index_points(pcs, tgt).contiguous()


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)


state, jv_state, prev_pts, prev_cols, k = {}, None, None, None, 0
print("refresh iter | moved>5mm moved>2cm | cols changed | cold ms | warm-race ms | jv re-solve ms (freed left steps) | per-wave re-solve | winners")
while i < n_iter:
    eng.peek_forward()
    src_pts = index_points(eng.pc_trans, src.expand(B, nf)).contiguous()
    cost = cdist(src_pts, tgt_pts)
    sample = k % EVERY == 0 and prev_pts is not None
    if sample:
        _, t_cold = timed(lambda: linear_sum_assignment_batch(cost, points=(src_pts, tgt_pts), race=True))
        js = {"prices": jv_state["prices"].clone(), "cols": jv_state["cols"].clone()}
        (a_jv, fb_jv, st_jv), t_jv = timed(lambda: linear_sum_assignment_points(src_pts, tgt_pts, js, return_stats="full", per_wave=False))
        jm = {"prices": jv_state["prices"].clone(), "cols": jv_state["cols"].clone()}
        (a_mw, fb_mw, st_mw), t_mw = timed(lambda: linear_sum_assignment_points(src_pts, tgt_pts, jm, return_stats="full", per_wave=True))
    (assign, fb, st), t_warm = timed(lambda: linear_sum_assignment_batch(cost, points=(src_pts, tgt_pts), race="warm", state=state,
                                                                          return_stats="full"))
    cols_np = np.stack([c for _, c in assign])
    if sample:
        d = (src_pts - prev_pts).norm(dim=-1)
        same = all(np.array_equal(a[1], c) for a, c in zip(a_jv, cols_np)) and all(np.array_equal(a[1], c) for a, c in zip(a_mw, cols_np))
        print(f"{k:5d} {i:6d} | {(d > 0.005).sum().item() / B:7.1f} {(d > 0.02).sum().item() / B:7.1f} | "
              f"{(cols_np != prev_cols).sum() / B:7.1f} | {t_cold:7.2f} | {t_warm:7.2f} | {t_jv:7.2f} "
              f"({(st_jv[:, 0] & 0xffff).mean():.0f} {st_jv[:, 1].mean():.0f} {st_jv[:, 2].mean():.0f} max {st_jv[:, 2].max()}) | "
              f"mw {t_mw:6.2f} (left {st_mw[:, 1].mean():.0f} steps {st_mw[:, 2].mean():.0f} max {st_mw[:, 2].max()} arr {(st_mw[:, 3] >> 8).mean():.0f} "
              f"conf {jm.get('commit_conflicts', np.zeros(1)).mean():.0f}) fb {fb_mw} | "
              f"{(st[:, 0] >> 16).tolist()} fb {fb} {fb_jv} same {same}", flush=True)
    jv_state = {"prices": state["prices"].clone(), "cols": state["cols"].clone()}
    prev_pts, prev_cols = src_pts, cols_np
    cols = torch.from_numpy(cols_np).to(dev)
    eng.set_assignment(src[0], tgt.gather(1, cols), 0.3)
    eng.step(5); i += 5; k += 1
