#!/usr/bin/env python3
"""Cold solves of the assignment problems the tail (19 x 4096^2, compute_ass_err) and the kinematic loop's first refresh
(19 x 2048^2) pose: time, phases, rounds, bids per matrix."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reart_amd.utils.lap import cdist, linear_sum_assignment_batch

dev = torch.device("cuda:0")
T, N, cano_idx = 20, 4096, 10
eng, seq, model = bench.build_instance(dev, T, N, cano_idx, seed=2)
eng.capture(50); eng.step(2000); torch.cuda.synchronize()
cano, pcs = eng.caller_clouds()
with torch.no_grad():
    pred, seg0, trans0 = model(cano)
import ctypes
from reart_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
names = ["matrix max + init", "phase start: release scan = first bids", "free-row list", "bids (2+ bidders)", "single-bidder chains",
         "resolution", "certificate + outputs"]
for n in [int(x) for x in os.environ.get("LAP_NS", "4096,2048").split(",")]:
    pa, pb = pred[:, :n].contiguous(), pcs[:, :n].contiguous()
    cost = cdist(pa, pb)
    use_points = os.environ.get("LAP_POINTS", "1") != "0"
    race = os.environ.get("LAP_RACE", "0") == "1"
    for rep in range(2):
        if hasattr(lib, "reart_debug_auction_phase"):
            lib.reart_debug_auction_phase((ctypes.c_ulonglong * 320)(), 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, fb, st = linear_sum_assignment_batch(cost, return_stats="full", points=(pa, pb) if use_points else None, race=race)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
    print(f"n={n}: {ms:.1f} ms, fallbacks {fb}; per matrix: phases {(st[:,0] & 0xffff).mean():.1f}, rounds {st[:,1].mean():.0f} (max {st[:,1].max()}), "
          f"bids {st[:,2].mean():.0f} (max {st[:,2].max()}), bids/round {st[:,2].sum()/st[:,1].sum():.1f}, certificate rounds {st[:,3].mean():.1f}")
    if hasattr(lib, "reart_debug_auction_phase"):
        buf = (ctypes.c_ulonglong * 320)()
        lib.reart_debug_auction_phase(buf, 0)
        v = np.array(list(buf), dtype=np.float64).reshape(32, 10)[:cost.shape[0]]
        slow = int(v.sum(1).argmax())
        print(f"   slowest workgroup {slow}: {v[slow].sum() / 1e5:.1f} ms (s_memtime at 100 MHz)")
        print(f"   single-bidder chains: {v[slow, 7]:.0f} with {v[slow, 8]:.0f} links in the slowest workgroup, mean {v[:, 7].mean():.0f} with {v[:, 8].mean():.0f}")
        if hasattr(lib, "reart_debug_auction_trace"):
            tb = (ctypes.c_int * 128)()
            lib.reart_debug_auction_trace(tb)
            tr = np.array(list(tb)).reshape(32, 4)[:int(st[0, 0])]
            print("   workgroup 0 per phase (released | rounds | bids incl. links / search steps | search steps):")
            print("   " + "  ".join(f"{a}|{b}|{c}|{d}" for a, b, c, d in tr))
        for k, nm in enumerate(names):
            print(f"   {nm:42s} {v[slow, k] / 1e5:8.2f} ms   (mean over workgroups {v[:, k].mean() / 1e5:8.2f})")
