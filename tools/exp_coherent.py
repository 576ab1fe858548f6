import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, time
from reart_amd.networks.model import BaseModel
from reart_amd.relax import RelaxEngine
from reart_amd.synthetic import make_sequence, split_canonical
dev = torch.device("cuda:0")
seq = make_sequence(T=20, n_parts=8, pts_per_part=512, seed=2, n_ref=3000)
cano, pcs = split_canonical(seq["complete"], 10)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for name, kw in (("tau schedule (random parts)", {}), ("fixed_tau=0.02 (coherent parts)", dict(fixed_tau=0.02))):
    torch.manual_seed(2)
    model = BaseModel(num_parts=20, pose_len=19).to(dev)
    eng = RelaxEngine(t(cano), t(pcs), model, 10, [t(r) for r in seq["ref_loc"]], [t(f) for f in seq["ref_flow"]], seed=2, **kw)
    eng.step(300)
    ph = eng.step_timed(20)
    print(name, {k: round(v * 1e3, 1) for k, v in ph.items()}, "loss", eng.last_losses().cpu().numpy())
