#!/usr/bin/env python3
"""Launch by launch through the assignment re-solves of a rocprofv3 kernel trace: for every position of the solve's launch
sequence (first pass kernel ... tie check, copies and fills included) the kernel's name, its mean duration and the mean idle
time in front of it -- which launches a solve waits for without computing.  usage: solve_gaps.py trace.csv [--last N]"""
import csv, sys
from collections import defaultdict
import numpy as np
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
# a solve: from a pass kernel that follows a tie check / certificate (or the trace's start) to the next such pass kernel
starts = []
passes = 2
for k, (s, e, n) in enumerate(rows):
    if "lap_jv_pass" in n:
        if passes >= 2:
            starts.append(k); passes = 0
        passes += 1
starts = starts[-last - 1:] if last else starts
seqs = defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    names = tuple(n.split("(")[0][:48] for _, _, n in rows[a:b])
    seqs[names].append((a, b))
names, spans = max(seqs.items(), key=lambda kv: len(kv[1]))
print(f"{len(starts) - 1} solve periods, {len(seqs)} distinct launch sequences; the commonest ({len(spans)} periods, {len(names)} launches):")
dur = np.zeros((len(spans), len(names))); gap = np.zeros_like(dur)
for i, (a, b) in enumerate(spans):
    for k in range(a, b):
        dur[i, k - a] = rows[k][1] - rows[k][0]
        gap[i, k - a] = rows[k][0] - rows[k - 1][1] if k > 0 else 0
print(f"{'#':>3} {'kernel':48} {'idle before (us) p50':>22} {'mean':>8} {'duration p50':>14} {'mean':>8}")
for k, n in enumerate(names):
    print(f"{k:3d} {n:48} {np.median(gap[:, k]) / 1e3:22.1f} {gap[:, k].mean() / 1e3:8.1f} {np.median(dur[:, k]) / 1e3:14.1f} {dur[:, k].mean() / 1e3:8.1f}")
print(f"    period p50 {np.median((dur + gap).sum(1)) / 1e3:.1f} us = kernels {np.median(dur.sum(1)) / 1e3:.1f} + idle {np.median(gap.sum(1)) / 1e3:.1f}")
