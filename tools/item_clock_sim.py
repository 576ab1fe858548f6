#!/usr/bin/env python3
"""Greedy list-scheduling simulation of one search launch from measured item lifetimes (clock build):
launch order vs longest-first.   make -C reart_amd/csrc clock && REART_LIB=reart_amd/csrc/libreart_hip_clock.so python tools/item_clock_sim.py"""
import ctypes, heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from reart_amd import _lib
eng, seq, model = bench.build_instance(torch.device("cuda:0"), 20, 4096, 10, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 14592
buf = (ctypes.c_ulonglong * (3 * n))()
def sim(lives, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for l in lives:
        t = heapq.heappop(h) + l
        end = max(end, t)
        heapq.heappush(h, t)
    return end
for it in (330, 3000):
    eng.step(it - int(eng.iter.item())); torch.cuda.synchronize()
    lib.reart_debug_item_clock(buf, n)
    v = np.array(list(buf), dtype=np.float64).reshape(n, 3)
    life = v[:, 1] - v[:, 0]
    prev = life.copy()
    eng.step(1); torch.cuda.synchronize()
    lib.reart_debug_item_clock(buf, n)
    v = np.array(list(buf), dtype=np.float64).reshape(n, 3)
    life = v[:, 1] - v[:, 0]
    print(f"iteration {it}: item life mean {life.mean():.0f} max {life.max():.0f}; correlation with the previous iteration's life {np.corrcoef(prev, life)[0, 1]:.2f}")
    for slots in (3072, 4096):
        fifo = sim(life, slots); lpt = sim(life[np.argsort(-life)], slots); lpt_prev = sim(life[np.argsort(-prev)], slots)
        rng = np.random.default_rng(0)
        rnd = sim(life[rng.permutation(n)], slots); rev = sim(life[::-1], slots)
        # mean life per tenth of the launch order (structure of the order)
        if slots == 4096:
            print("   mean life per tenth of the launch order:", [int(x.mean()) for x in np.array_split(life, 10)])
        print(f"   {slots} slots: random order {rnd:.0f}  reversed {rev:.0f}")
        print(f"   {slots} slots: ideal {life.sum() / slots:.0f}  launch order {fifo:.0f}  longest-first (oracle) {lpt:.0f}  longest-first by previous iteration {lpt_prev:.0f}  ticks")
