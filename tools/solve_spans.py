#!/usr/bin/env python3
"""GPU-side span of every assignment re-solve in a rocprofv3 kernel trace (from the first pass kernel's start to the end of the
last kernel of the solve: certificate or tie check), the kernels' summed time inside it and the gap to the next solve's start --
what part of `ms_per_solve` is kernels, what part launch gaps, what part host.  usage: solve_spans.py trace.csv [--last N]"""
import csv, sys
import numpy as np
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
solves, cur = [], None
for s, e, n in rows:
    if "lap_jv_pass" in n and (cur is None or cur["passes"] >= 2):
        cur = {"t0": s, "t1": e, "busy": 0, "passes": 0, "n": 0}
        solves.append(cur)
    if cur is None:
        continue
    if "lap_" in n or "fillBuffer" in n:
        if s - cur["t1"] > 2_000_000:      # (a kernel far behind the solve belongs to something else)
            continue
        cur["t1"] = max(cur["t1"], e); cur["busy"] += e - s; cur["n"] += 1
        if "lap_jv_pass" in n:
            cur["passes"] += 1
sel = solves[-last:] if last else solves
span = np.array([c["t1"] - c["t0"] for c in sel]) / 1e3
busy = np.array([c["busy"] for c in sel]) / 1e3
nk = np.array([c["n"] for c in sel])
period = np.diff([c["t0"] for c in sel]) / 1e3
print(f"{len(sel)} solves: GPU span mean {span.mean():.1f} us (p50 {np.median(span):.1f}), kernels inside {busy.mean():.1f} us in {nk.mean():.1f} launches "
      f"(gaps inside the span {span.mean() - busy.mean():.1f} us); start-to-start period mean {period.mean():.1f} us p50 {np.median(period):.1f}")
