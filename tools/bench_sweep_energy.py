#!/usr/bin/env python3
"""Sweep over canonical indices WITH the end-of-run energy (what the reference's model selection needs): wall time of
6 instances (T = 20, N = 4096, ITERS iterations each), three in flight per GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reart_amd import sweep

dev = torch.device("cuda:0")
iters = int(os.environ.get("ITERS", 3000))
def make_engine(spec):
    eng, _, _ = bench.build_instance(dev, 20, 4096, spec["cano_idx"], 2, n_iter=iters)
    return eng
inst = [{"cano_idx": c} for c in (3, 6, 9, 12, 15, 18)]
sweep.run_sweep_engines(inst[:3], make_engine, 200, dev, per_gpu=3, chunk=100, energy=True)   # warm-up
for energy, overlap in ((False, False), (True, False), (True, True)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec, best = sweep.run_sweep_engines(inst, make_engine, iters, dev, per_gpu=3, chunk=100, energy=energy, overlap_tails=overlap)
    torch.cuda.synchronize()
    print(f"energy={energy} overlap_tails={overlap}: {time.perf_counter() - t0:.2f} s for {len(inst)} instances x {iters} iterations; best cano_idx {inst[best]['cano_idx']}"
          + (f"; total_err {rec[:, sweep.E_TOTAL].cpu().numpy().round(4).tolist()}" if energy else ""))
