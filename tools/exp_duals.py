#!/usr/bin/env python3
"""Does the CHOICE of optimal duals kept between refreshes matter for the re-solve?  The shortest-path solver leaves the
lowest feasible prices (every raise is the least possible); one Jacobi pass p_s(i) += theta * (v2_i - v1_i) (the slack of row
i between its column and its second choice; simultaneous raises only add slack elsewhere) moves them towards the auction's
end of the dual-optimal set.  Same engine trajectory, one kept state per theta, the one-search-at-a-time solver without a
race (clean statistics).  Usage: gpurun -- python tools/exp_duals.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd.networks.model import BaseModel
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.relax import RelaxEngine
from reart_amd.utils import lap

dev = torch.device("cuda:0")
from reart_amd.data import load_nao_demo
g = load_nao_demo()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cano, pcs, cano_idx = t(g["cano"]), t(g["pc_list"]), int(g["cano_idx"])
gt_pos = t(g["complete_gt_pc_list"])
rng = np.random.default_rng(0)
sel = [torch.from_numpy(rng.permutation(gt_pos.shape[1])[:3000]).to(dev) for _ in range(pcs.shape[0])]
refs = [gt_pos[k][s] for k, s in enumerate(sel)]
flows = [t(g["gt_flow_list"][k])[s] for k, s in enumerate(sel)]
torch.manual_seed(2)
model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
eng = RelaxEngine(cano, pcs, model, cano_idx, refs, flows, n_iter=15000, seed=2)
i = eng.capture(steps_per_graph=10)
eng.step(int(os.environ.get("START", 8000)) - i)
B, N = pcs.shape[:2]; nf = N // 4
zero = torch.zeros(1, dtype=torch.long, device=dev)
src = farthest_point_sample(cano[None], nf, start=zero, cuda_mode=True)
tgt = farthest_point_sample(pcs, nf, start=zero.expand(B), cuda_mode=True)
tgt_pts = index_points(pcs, tgt).contiguous()
thetas = [float(x) for x in os.environ.get("THETAS", "0,0.25,0.5,0.75,1.0").split(",")]
states = {th: None for th in thetas}
lap.RESOLVE_RACERS = int(os.environ.get("RACERS", 1))


def centre(state, src_pts, th):
    """p_s(i) += th * (v2_i - v1_i) for every row, all at once."""
    if th == 0:
        return
    vals = lap.cdist(src_pts, tgt_pts).double() + state["prices"][:, None, :]
    two = torch.topk(vals, 2, dim=2, largest=False).values
    cols = state["cols"].long()
    cur = torch.gather(vals, 2, cols[:, :, None])[:, :, 0]
    slack = (two[:, :, 1] - torch.maximum(cur, two[:, :, 0])).clamp_min(0)
    state["prices"].scatter_add_(1, cols, th * slack)


for k in range(int(os.environ.get("REPS", 8))):
    eng.peek_forward()
    src_pts = index_points(eng.pc_trans, src.expand(B, nf)).contiguous()
    line = f"refresh {k}:"
    out = None
    for th in thetas:
        if states[th] is None:
            states[th] = {}
            out = lap.linear_sum_assignment_points(src_pts, tgt_pts, states[th])
        else:
            st_ = states[th]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o, fb, st = lap.linear_sum_assignment_points(src_pts, tgt_pts, st_, return_stats="full", per_wave=False)
            torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
            if out is not None:
                assert all(np.array_equal(a[1], b_[1]) for a, b_ in zip(o, out)), "assignments differ"
            out = o
            line += (f" | th {th}: {ms:6.2f} ms rel {(st[:, 0] & 0xffff).mean():.0f} left {st[:, 1].mean():.0f} steps {st[:, 2].mean():.0f} "
                     f"max {st[:, 2].max()} arr {(st[:, 3] >> 8).mean():.0f} fb {fb}")
        centre(states[th], src_pts, th)
    print(line, flush=True)
    cols = torch.from_numpy(np.stack([c for _, c in out])).to(dev)
    eng.set_assignment(src[0], tgt.gather(1, cols), 0.3)
    eng.step(5)
