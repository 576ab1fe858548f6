#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection CSV rows per (kernel, counter):  python tools/pmc_sum.py DIR [substr]"""
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if len(sys.argv) > 2 and sys.argv[2] not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()): print(f"    {c:32s} {x:18.0f}  /dispatch {x / n[(k, c)]:14.1f}  (n={n[(k, c)]})")
