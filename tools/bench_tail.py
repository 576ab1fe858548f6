#!/usr/bin/env python3
"""End of an instance at the headline size (T = 20, N = 4096, P = 20): structure extraction and the model-selection
energy after a short optimisation; wall time of each stage (device-synchronised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reart_amd import tail
from reart_amd.utils import graph_utils as gu
from reart_amd.utils.model_utils import compute_ass_err, compute_group_temporal_err, compute_pc_transform

dev = torch.device("cuda:0")
iters = int(os.environ.get("ITERS", 3000))
eng, seq, model = bench.build_instance(dev, 20, 4096, 10, 2, n_iter=iters)
eng.capture(steps_per_graph=10)
eng.step(iters); torch.cuda.synchronize()
cano, pcs = eng.cano, eng.pc_list

def timed(name, fn, reps=3):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / reps * 1e3:9.2f} ms")
    return out

with torch.no_grad():
    _, seg0, trans0 = model(cano)
trans0 = trans0.detach()
seg, trans, conn = timed("extract_structure (whole)", lambda: tail.extract_structure(seg0, trans0, cano))
print("parts", trans.shape[1], "edges", conn.tolist())
from reart_amd.knn_cuda import KNN
knn = KNN(k=1, transpose_mode=True)
dn = timed("  denoise_seg_label", lambda: gu.denoise_seg_label(seg0.clone(), cano, knn, min_num=20))
mg = timed("  merging_wrapper", lambda: gu.merging_wrapper(dn, trans0, cano, None, 3e-2, n_it=2))
timed("  mst_wrapper", lambda: gu.mst_wrapper(mg, trans0, cano, None))
uni = torch.unique(mg)
timed("    fps_sample_cano (kernel+gather)", lambda: gu.fps_sample_cano(cano, mg, uni, 20))
P = trans0.shape[1]; ar = torch.arange(P, device=dev)
pairs = torch.stack([ar.repeat_interleave(P), ar.repeat(P)], 1)
timed("    screw_fit 400 pairs x 19", lambda: gu.screw_fit(trans0, pairs))
pred = compute_pc_transform(cano, trans, seg)
timed("energy_terms (whole)", lambda: tail.energy_terms(cano, pcs, seg, trans, conn, 10, pred), reps=1)
timed("  compute_ass_err 19 x 4096^2", lambda: compute_ass_err(pred, pcs), reps=1)
timed("  compute_screw_cost", lambda: gu.compute_screw_cost(trans, conn))
comp = torch.cat((pred[:10], cano[None], pred[10:]))
timed("  compute_group_temporal_err", lambda: compute_group_temporal_err(comp, seg))
timed("snapshot_metrics (cd_err)", lambda: tail.snapshot_metrics(cano, pcs, seg, trans, 10))
from reart_amd.utils.lap import linear_sum_assignment_batch
cost = torch.cdist(pred, pcs)
out, fb, st = linear_sum_assignment_batch(cost, return_stats="full")
print("LAP stats per matrix (phases, rounds, bids, certificate rounds): mean", st.mean(0), "max", st.max(0), "fallbacks", fb)
