#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 kernel trace restricted to a WINDOW of dispatches: `--last N` keeps, per kernel
name, only its last N dispatches (the timed region of a bench run comes last: warm-up launches, cold solves and set-up
launches before it would otherwise be averaged in -- VERDICT r03 weak #6: the kinematic profile averaged nine warm-up
re-solves into the re-solve kernel's mean).  Input: the trace as rocprofv3 writes it, either the SQLite file
(`*_results.db`, table/view `kernels`) or the CSV (`*_kernel_trace.csv`).  Output: CSV rows like rocprofv3's kernel_stats
(Name, Calls, TotalDurationNs, AverageNs, MinNs, MaxNs, Percentage), over the window.
    python tools/kernel_window_stats.py gpurun_out/prof/x_results.db --last 100 --match lap_ > profiles/r04_kernel_stats_kinematic.csv"""
import argparse
import csv
import sqlite3
import sys
from collections import defaultdict


def load(path):
    """-> [(name, start_ns, end_ns)] in dispatch order."""
    if path.endswith(".db"):
        db = sqlite3.connect(path)
        rows = db.execute("select name, start, end from kernels order by start").fetchall()
        return [(r[0], int(r[1]), int(r[2])) for r in rows]
    out = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out.sort(key=lambda x: x[1])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last", type=int, default=0, help="keep the last N dispatches of every kernel (0: all)")
    ap.add_argument("--match", default="", help="only kernels whose name contains this")
    a = ap.parse_args()
    per = defaultdict(list)
    for name, s, e in load(a.trace):
        if a.match in name:
            per[name].append(e - s)
    rows = []
    for name, d in per.items():
        w = d[-a.last:] if a.last else d
        rows.append((name, len(w), sum(w), sum(w) / len(w), min(w), max(w)))
    tot = sum(r[2] for r in rows) or 1
    wr = csv.writer(sys.stdout)
    wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "Window"])
    for r in sorted(rows, key=lambda r: -r[2]):
        wr.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100 * r[2] / tot, 3), f"last {a.last}" if a.last else "all"])


if __name__ == "__main__":
    main()
