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
    ap.add_argument("--after-last", default="", metavar="SUBSTR",
                    help="window by TIME: keep only dispatches that start after the last dispatch of a kernel whose name contains "
                         "SUBSTR (VERDICT r05 weak #8: a trace that holds the recipe AND the projection -- the recipe's n = 1024 kernels "
                         "are `<16, ...>` instances, so `--after-last 'lap_jvmw_kernel<16'` leaves exactly the projection)")
    ap.add_argument("--before-first", default="", metavar="SUBSTR", help="... and that start before the first dispatch (after that point) of such a kernel")
    ap.add_argument("--skip", type=int, default=0, help="drop the first N dispatches of every kernel inside the window (cold solve, warm-up)")
    ap.add_argument("--first", type=int, default=0, help="keep only the first N dispatches of every kernel inside the window (after --skip)")
    a = ap.parse_args()
    trace = load(a.trace)
    t_from = 0
    if a.after_last:
        ends = [e for name, s, e in trace if a.after_last in name]
        if not ends:
            sys.exit(f"no kernel matching {a.after_last!r} in the trace")
        t_from = max(ends)
    t_to = None
    if a.before_first:
        starts = [s for name, s, e in trace if a.before_first in name and s >= t_from]
        t_to = min(starts) if starts else None
    per = defaultdict(list)
    for name, s, e in trace:
        if a.match in name and s >= t_from and (t_to is None or s < t_to):
            per[name].append(e - s)
    rows = []
    for name, d in per.items():
        w = d[a.skip:]
        w = w[:a.first] if a.first else w
        w = w[-a.last:] if a.last else w
        if not w:
            continue
        rows.append((name, len(w), sum(w), sum(w) / len(w), min(w), max(w)))
    tot = sum(r[2] for r in rows) or 1
    wr = csv.writer(sys.stdout)
    wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "Window"])
    for r in sorted(rows, key=lambda r: -r[2]):
        window = " ".join(x for x in (f"after last {a.after_last}" if a.after_last else "", f"before first {a.before_first}" if a.before_first else "",
                                      f"skip {a.skip}" if a.skip else "", f"first {a.first}" if a.first else "", f"last {a.last}" if a.last else "") if x)
        wr.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100 * r[2] / tot, 3), window or "all"])


if __name__ == "__main__":
    main()
