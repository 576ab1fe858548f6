import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bench import build_instance
dev = torch.device("cuda:0")
use_flow = "--no-flow" not in sys.argv
eng, seq, model = build_instance(dev, 20, 4096, 10, seed=2, use_flow=use_flow)
for i in range(60):
    eng.step(1)
    it, log = eng.loss_log()
    r = log[-1].cpu().numpy()
    pt = eng.pc_trans
    print(i, r, float(pt.abs().max()), bool(torch.isfinite(pt).all()), float(model.proposal_6d.abs().max()), float(model.proposal_t.abs().max()), float(model.seg_head.model[2].weight.abs().max()))
    if not np.isfinite(r).all(): break
print(eng.step_timed(3))
