#!/bin/bash
# Per-launch durations of ONE extractor forward, in launch order (rocprofv3 kernel trace of bench.py --config extractor).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/ext_trace
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/t -- python3 bench.py --config extractor --steps 3 --warmup 2 > $O/bench.json 2> $O/err.txt
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last forward: find the last fps_kernel<..> group start
idx = [i for i, n in enumerate(names) if "fps_kernel" in n]
# a forward has 2 FPS launches per call x 2 calls? print the tail after the 4th-from-last fps launch
start = idx[-2] if len(idx) >= 2 else 0
tot = small = nsmall = 0
for r in rows[start:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d < 8.0:
        small += d; nsmall += 1
        continue
    print(f"{d:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>9}  {r['Kernel_Name'][:100]}")
print(f"launches under 8 us: {nsmall}, {small:.1f} us in total")
wall = (int(rows[-1]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])) / 1e3
print(f"sum of durations {tot:.1f} us, first start to last end {wall:.1f} us")
PY
rm -rf $O/t
