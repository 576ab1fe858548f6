#!/usr/bin/env python3
"""it/s of the headline step for every canonical index a multi-GPU sweep hands out (rank r -> cano (T//2 + r) % T):
how much the per-rank instances of `bench.py --gpus N` differ in cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
T, N = 20, 4096
for r in range(8):
    c = (T // 2 + r) % T
    eng, _, _ = bench.build_instance(dev, T, N, c, seed=2 + r)
    used = eng.capture(steps_per_graph=50)
    eng.step(150 - used)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.step(1500)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rank {r} cano_idx {c}: {1500 / dt:8.1f} it/s", flush=True)
    del eng
