#!/usr/bin/env python3
"""What a kinematic-projection iteration does BETWEEN two assignment re-solves (the forward, the flow blends, the losses' gradients,
the FK backward, Adam, the next forward), from a rocprofv3 kernel trace: wall time from the end of a solve's last kernel to the next
solve's first pass kernel, and the kernels inside it.   usage: iteration_glue.py kernel_trace.csv [iterations from the end]"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ends = [i for i, (s, e, n) in enumerate(rows) if "lap_tie_cycle" in n or ("lap_jv_kernel" in n and ", 2>" in n)]
# one end per solve: the LAST of the two names when both occur
ends = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] != i + 1 and not any("lap_tie_cycle" in rows[j][2] for j in range(i + 1, min(i + 4, len(rows))))]
idx = ends[-last:]
per, cnt, wall, n = collections.defaultdict(float), collections.defaultdict(int), 0.0, 0
for a in idx[:-1]:
    t0, j = rows[a][1], a + 1
    while j < len(rows) and "lap_jv_pass" not in rows[j][2]:
        s, e, name = rows[j]
        key = name.split("(")[0][:70]
        per[key] += e - s; cnt[key] += 1; j += 1
    if j < len(rows):
        wall += rows[j][0] - t0; n += 1
print(f"{n} iterations: from a solve's end to the next solve's first kernel {wall / n / 1e3:.1f} us of wall per iteration; kernels inside (us per iteration, launches per iteration):")
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  {k:72s} {v / n / 1e3:7.1f}  x{cnt[k] / n:.1f}")
print(f"  sum of kernel time {sum(per.values()) / n / 1e3:.1f} us in {sum(cnt.values()) / n:.1f} launches")
