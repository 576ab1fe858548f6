#!/usr/bin/env python3
"""Print the kernel timeline of a few iterations from a rocprofv3 --kernel-trace CSV:  python tools/timeline.py DIR [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
keys = ("knn_", "base_", "flow_blend", "chamfer_grad", "bookkeep")
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in keys)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
seq = rows[len(rows) // 2: len(rows) // 2 + n]
t0 = int(seq[0]["Start_Timestamp"])
for r in seq:
    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{r['Kernel_Name'][:44]:44s} start {st / 1e3:8.1f}  end {en / 1e3:8.1f}  dur {(en - st) / 1e3:6.1f} us  queue {r.get('Queue_Id', '')}")
