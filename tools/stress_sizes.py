"""Robustness sweep of the fused step over problem sizes (no oracle: finite results, use_boxes on/off agree on the
losses to fp32 round-off, nothing crashes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd.networks.model import BaseModel
from reart_amd.relax import RelaxEngine
from reart_amd.synthetic import make_sequence, split_canonical
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for (T, parts, ppp, P, nref, flow) in ((3, 1, 70, 20, 64, True), (10, 8, 1024, 20, 5000, True), (6, 4, 2500, 10, 3000, True),
                                       (40, 8, 128, 8, 500, True), (20, 8, 512, 32, 3000, False), (4, 2, 33, 3, 40, True)):
    seq = make_sequence(T=T, n_parts=parts, pts_per_part=ppp, seed=1, n_ref=nref, with_flow=flow)
    cano, pcs = split_canonical(seq["complete"], T // 2)
    out = []
    for sort in (True, False):
        torch.manual_seed(0)
        model = BaseModel(num_parts=P, pose_len=T - 1).to(dev)
        try:
            eng = RelaxEngine(t(cano), t(pcs), model, T // 2, [t(r) for r in seq["ref_loc"]] if flow else None,
                              [t(f) for f in seq["ref_flow"]] if flow else None, n_iter=100, spatial_sort=sort)
            eng.capture(); eng.step(19)
            it, log = eng.loss_log()
            out.append(log[-1].cpu().numpy())
        except Exception as e:
            out.append(repr(e)[:100])
    print(f"T={T} N={parts*ppp} P={P} flow={flow}:", out[0], "| unsorted:", out[1])
