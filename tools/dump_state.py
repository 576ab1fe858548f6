#!/usr/bin/env python3
"""Dump the search problem of the bench instance at a few iterations (for offline filter experiments, tools/sim_filter.py):
the stored (k-d order) clouds of two consecutive iterations -- the exact neighbours of the first are the seeds of the second.
    python tools/dump_state.py gpurun_out/state.npz 300 1500"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench

out, iters = sys.argv[1], [int(v) for v in sys.argv[2:]] or [300, 1500]
dev = torch.device("cuda:0")
eng, seq, model = bench.build_instance(dev, 20, 4096, 10, 2)
d = {"cano": eng.cano.cpu().numpy(), "pc_list": eng.pc_list.cpu().numpy(), "ref_loc": eng.ref_loc.cpu().numpy(),
     "ref_off": eng.ref_off.cpu().numpy(), "cano_idx": np.int64(10)}
def sampled(eng):
    """the part the Gumbel sampler drew for every point (not seg_part, the arg-max of the logits): the one whose transforms
    reproduce pc_trans in every frame"""
    T = eng.trans_list                                            # [B,P,4,4]
    cand = torch.einsum("bpij,nj->bpni", T[:, :, :3, :3], eng.cano) + T[:, :, None, :3, 3]
    err = ((cand - eng._pc_trans[:, None]) ** 2).sum(-1).sum(0)   # [P,N]
    assert float(err.min(0).values.max()) < 1e-8
    return err.argmin(0).cpu().numpy()

for it in iters:
    eng.step(it - 1 - int(eng.iter.item()))
    torch.cuda.synchronize()
    eng.step(1); torch.cuda.synchronize()
    d[f"prev_{it}"] = eng._pc_trans.cpu().numpy().copy()          # output of iteration it - 1 (stored order)
    d[f"segprev_{it}"] = sampled(eng)
    eng.step(1); torch.cuda.synchronize()
    d[f"cur_{it}"] = eng._pc_trans.cpu().numpy().copy()
    d[f"seg_{it}"] = sampled(eng)                                 # the SAMPLED part of every point in that iteration (stored order)
np.savez_compressed(out, **d)
print("wrote", out, {k: v.shape for k, v in d.items()})
