#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table (markdown)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for n, c, t, a, p in rows:
    if len(n) > 90: n = n[:87] + "..."
    print(f"| `{n}` | {c} | {t:.1f} | {a:.3f} | {p:.2f} |")
