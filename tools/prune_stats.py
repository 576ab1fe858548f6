#!/usr/bin/env python3
"""Filter statistics and throughput of the pruned search along an optimisation run.

    make -C reart_amd/csrc stats
    REART_LIB=reart_amd/csrc/libreart_hip_stats.so python tools/prune_stats.py        # counters
    python tools/prune_stats.py --no-stats                                            # throughput only

Prints, per window of the run: iterations/s and (stats build) per search the mean number of boxes
that pass the coarse filter / are scanned, out of the boxes of one target cloud.
"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reart_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=15000)
ap.add_argument("--window", type=int, default=1000)
ap.add_argument("--probe", type=int, default=100, help="iterations measured at the start of each window")
ap.add_argument("--no-stats", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
eng, seq, model = bench.build_instance(dev, 20, 4096, 0, 2, n_iter=args.iters)
lib = ctypes.CDLL(_lib.LIB_PATH)
have = hasattr(lib, "reart_debug_prune_stats") and not args.no_stats
buf = (ctypes.c_ulonglong * 8)()
eng.capture()
done = 0
print("iter  it/s   | K=1: coarse-pass scanned (of 256 boxes per cloud, per wave); scanned boxes needed by 1 / 2-3 / 4-7 / 8+ of the 64 queries | recon flow")
while done < args.iters:
    if have:
        lib.reart_debug_prune_stats(buf, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.step(args.probe)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    line = f"{done:6d} {args.probe / dt:8.1f}"
    if have:
        lib.reart_debug_prune_stats(buf, 1)
        v = list(buf)
        n1 = args.probe * 2 * 19 * 64    # (wave, cloud) pairs per probe; slices share the boxes of a cloud
        n3 = args.probe * 19 * 64
        h = max(sum(v[4:8]), 1)
        line += f" | {v[1] / n1:7.1f} {v[2] / n1:7.1f}; {v[4] / h:.0%} {v[5] / h:.0%} {v[6] / h:.0%} {v[7] / h:.0%}"
    l = eng.last_losses()
    line += f" | {l}"
    print(line, flush=True)
    rest = min(args.window - args.probe, args.iters - done - args.probe)
    if rest > 0:
        eng.step(rest)
    done += args.window
