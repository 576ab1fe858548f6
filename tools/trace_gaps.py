#!/usr/bin/env python3
"""GPU idle time between consecutive kernels of a rocprofv3 kernel trace (csv), attributed to the kernel that FOLLOWS the gap:
where a host-driven loop leaves the device waiting.  `--last N` keeps the last N dispatches (the timed window).
    python tools/trace_gaps.py <kernel_trace.csv> --last 40000"""
import argparse, csv, collections

ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--last", type=int, default=0); ap.add_argument("--top", type=int, default=14)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
if a.last:
    rows = rows[-a.last:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gap = collections.Counter(); cnt = collections.Counter(); dur = collections.Counter()
end = rows[0][1]
for s, e, k in rows[1:]:
    g = max(0, s - end)
    gap[k] += g; cnt[k] += 1; dur[k] += e - s
    end = max(end, e)
print(f"dispatches {len(rows)}, span {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms, idle {sum(gap.values()) / 1e6:.2f} ms")
print("kernel,calls,avg_us,idle_before_avg_us,idle_before_total_ms")
for k, g in gap.most_common(a.top):
    print(f"{k},{cnt[k]},{dur[k] / cnt[k] / 1e3:.2f},{g / cnt[k] / 1e3:.2f},{g / 1e6:.2f}")
