#!/usr/bin/env python3
"""compute_ass_err at full size (the model-selection energy, reference utils/model_utils.py:92-104): (T-1) matrices of
4096 x 4096 -- GPU auction + certificate vs scipy on the host cores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.optimize import linear_sum_assignment
from reart_amd.utils.lap import linear_sum_assignment_batch
from reart_amd.utils.model_utils import compute_ass_err

dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "structure.npz"))
pred, pcs = torch.from_numpy(g["pred"]).to(dev), torch.from_numpy(g["pc_list"]).to(dev)
nb = int(os.environ.get("NB", pred.shape[0]))
pred, pcs = pred[:nb].contiguous(), pcs[:nb].contiguous()
cost = torch.cdist(pred, pcs).contiguous()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, fb, st = linear_sum_assignment_batch(cost, return_stats="full")
    torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
print("per matrix: phases, auction rounds, bids, certificate rounds\n", st)
print(f"{nb} x 4096^2: GPU {t_gpu*1e3:.0f} ms, host fallbacks {fb}")
t0 = time.perf_counter(); e = compute_ass_err(pred, pcs); torch.cuda.synchronize()
print(f"compute_ass_err {float(e):.9f} in {(time.perf_counter()-t0)*1e3:.0f} ms; golden (all 9 frames) {float(g['ass_err']):.9f}")
if os.environ.get("SCIPY", "1") == "1":
    ch = cost[:2].cpu().numpy()
    t0 = time.perf_counter(); ref = [linear_sum_assignment(c) for c in ch]; t_cpu = (time.perf_counter() - t0) / 2
    same = [bool(np.array_equal(out[b][1], ref[b][1])) for b in range(2)]
    print(f"scipy: {t_cpu*1e3:.0f} ms per matrix (serial); identical permutation on the first two: {same}")
