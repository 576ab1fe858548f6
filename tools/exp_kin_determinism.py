#!/usr/bin/env python3
"""Is the kinematic projection loop deterministic run to run, and do the flow blends on side streams change its results?
200 iterations of KinematicEngine (README.md:125 configuration) from the reference's kinematic-2 checkpoint on perturbed frames:
runs A and B with one blend stream, run C with six -- parameters compared bit for bit every 50 iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_kinematic_engine_gpu import _model, G, t
from reart_amd.kinematic_engine import KinematicEngine

dev = torch.device("cuda:0")
k = np.load(os.path.join(G, "kinematic.npz"))
cano = t(k["cano_pc"], dev)
rng = np.random.default_rng(5)
B, N = 9, cano.shape[0]
with torch.no_grad():
    pcs = _model(dev, k, cano)(cano)[0]
pcs = (pcs + t(rng.normal(0, 0.004, (B, N, 3)).astype(np.float32), dev)).contiguous()
pcs = torch.stack([p[torch.from_numpy(rng.permutation(N)).to(dev)] for p in pcs])
comp = torch.cat((pcs[:2], cano[None], pcs[2:]), dim=0)
sel = [torch.from_numpy(rng.permutation(N)[:3000]).to(dev) for f in range(B)]
refs = [comp[f][s] for f, s in enumerate(sel)]
flows = [(comp[f + 1][s] - comp[f][s]) * 0.5 for f, s in enumerate(sel)]
snaps = {}
for tag, side in (("A", 1), ("B", 1), ("C", 6)):
    KinematicEngine.SIDE_STREAMS = side
    m = _model(dev, k, cano)
    eng = KinematicEngine(m, cano, pcs, 2, refs, flows, assign_iter=0, assign_gap=1, downsample=2)
    out = []
    cols_log, cost_log = [], []
    for i in range(200):
        eng.iteration(i)
        cols_log.append(eng.lap_state["cols"].clone())
        d_ = (eng._pc_src.double() - eng.matched.double()).norm(dim=-1)          # the matched pairs' distances, per problem in float64
        cost_log.append(d_.sum(1).clone())
        if (i + 1) % 50 == 0:
            out.append(torch.cat([getattr(m, n_).detach().reshape(-1).clone() for n_ in ("axis_list", "moment_list", "theta_list")]))
    snaps[tag] = out
    snaps[tag + "_cols"], snaps[tag + "_cost"] = cols_log, cost_log
    print(tag, "fallbacks", eng.lap_fallbacks)
for a, b in (("A", "B"), ("A", "C")):
    print(a, "vs", b, ["equal" if torch.equal(x, y) else f"max diff {float((x - y).abs().max()):.3e}" for x, y in zip(snaps[a], snaps[b])])

for a, b in (("A", "B"), ("A", "C")):
    for i, (x, y) in enumerate(zip(snaps[a + "_cols"], snaps[b + "_cols"])):
        if not torch.equal(x, y):
            bad = (x != y).any(1).nonzero().flatten().tolist()
            ca, cb = snaps[a + "_cost"][i], snaps[b + "_cost"][i]
            print(f"{a} vs {b}: first different assignment at iteration {i}, problems {bad}: rows that differ {[(int((x[p_] != y[p_]).sum())) for p_ in bad]}, "
                  f"total cost {[float(ca[p_]) for p_ in bad]} vs {[float(cb[p_]) for p_ in bad]}, relative difference "
                  f"{[abs(float(ca[p_] - cb[p_])) / float(ca[p_]) for p_ in bad]}")
            # were the INPUTS still identical at that iteration?  (the parameters of the iteration before)
            break
