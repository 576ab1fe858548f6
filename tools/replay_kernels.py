#!/usr/bin/env python3
"""Per-kernel time of every REPLAYED re-solve (tools/replay_tail.py under `rocprofv3 --kernel-trace`, RAW_OUT=prefix) next to the
solver's statistics of that solve: what does the backward growth (lap_mc_forest_kernel) cost a solve with six rows left, and what
do the searches behind it cost?  A solve = the dispatches from one `lap_jv_pass_kernel` (first pass) to the certificate launch
(`lap_jv_kernel<.., 2>`); the LAST pass over a dumped solve is taken (the first is the untimed warm-up).
    python tools/replay_kernels.py <kernel_trace.csv> <RAW_OUT prefix>.<dump>.npz"""
import csv, sys, collections
import numpy as np

trace, raw = sys.argv[1], np.load(sys.argv[2])
raws, reps = raw["raws"], int(raw["reps"])
rows = []
with open(trace, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: ("pass" if "lap_jv_pass" in n else "setup" if "lap_jvmw_kernel" in n and ", 1, " in n else "search" if "lap_jvmw_kernel" in n else
                   "arr" if "lap_mc_arr" in n else "trees" if "lap_mc_trees" in n else "forest" if "lap_mc_forest" in n else "tighten" if "lap_mc_tighten" in n
                   else "cert" if "lap_jv_kernel" in n else "ties" if "lap_tie" in n else None)
solves, cur = [], None
for s, e, n in rows:
    k = short(n)
    if k is None:
        continue
    if k == "pass" and (cur is None or "search" in cur):      # the first pass of a solve (the second comes after the searches)
        if cur is not None and "search" not in cur:
            pass
        if cur is None or "cert" in cur or "search" in cur and cur.get("_passes", 0) >= 2:
            cur = collections.defaultdict(float)
            cur["_t0"] = s
            solves.append(cur)
    if cur is None:
        continue
    cur[k] += (e - s) / 1e3
    if k == "pass":
        cur["_passes"] = cur.get("_passes", 0) + 1
    cur["_t1"] = e
S = raws.shape[0]
per = reps + 1
if len(solves) != S * per:
    print(f"warning: {len(solves)} solves in the trace, expected {S} x {per}", file=sys.stderr)
left = (raws[:, :, 1] & 0xffff)
freed = raws[:, :, 0] & 0xffff
steps = raws[:, :, 2]
back = (raws[:, :, 0] >> 21) & 0x3ff
keys = ("pass", "setup", "tighten", "arr", "trees", "forest", "search", "cert")
print("solve  rows_left(max/sum) released(max) search_steps(max) back_rounds(max) | us: " + " ".join(f"{k:>8s}" for k in keys) + "    sum   wall")
tot = collections.defaultdict(float)
order = np.argsort(left.max(1))
table = []
for s_ in order:
    if (s_ + 1) * per - 1 >= len(solves):
        continue
    d = solves[(s_ + 1) * per - 1]
    su = sum(d[k] for k in keys)
    table.append((int(left[s_].max()), int(left[s_].sum()), d))
    print(f"{s_:5d}  {left[s_].max():5d}/{left[s_].sum():5d} {freed[s_].max():8d} {steps[s_].max():10d} {back[s_].max():10d}      | " +
          " ".join(f"{d[k]:8.1f}" for k in keys) + f" {su:7.1f} {(d['_t1'] - d['_t0']) / 1e3:7.1f}")
for name, sel in (("rows left (slowest problem) <= 8", lambda t: t[0] <= 8), ("9 .. 24", lambda t: 8 < t[0] <= 24), ("> 24", lambda t: t[0] > 24)):
    g = [t for t in table if sel(t)]
    if g:
        print(f"{name}: {len(g)} solves, mean us: " + " ".join(f"{k} {np.mean([t[2][k] for t in g]):.0f}" for k in keys) +
              f" | wall {np.mean([(t[2]['_t1'] - t[2]['_t0']) / 1e3 for t in g]):.0f}")
