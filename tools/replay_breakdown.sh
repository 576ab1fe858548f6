#!/bin/bash
# Per-kernel durations and inter-kernel gaps of the FIRST replay of a 20-iteration graph against the sixth
# (tools/graph_first_replay.py under rocprofv3 --kernel-trace): where do the driver-shaped run's 4 us per iteration go?
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/replay
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/t -- python3 tools/graph_first_replay.py > $O/out.txt 2> $O/err.txt
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:40] for r in rows]
# the six replays are the last 6 x 20 x 5 step-kernel dispatches
step = [i for i, n in enumerate(names) if any(k in n for k in ("knn_group", "base_fwd", "base_bwd_block", "base_bwd_finalize", "post_kernel"))]
tail = step[-600:]
for rep in (0, 1, 5):
    idx = tail[rep * 100:(rep + 1) * 100]
    dur = collections.defaultdict(float); gap = 0.0
    for a, b in zip(idx[:-1], idx[1:]):
        gap += (int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
    for i in idx:
        dur[names[i]] += (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    wall = (int(rows[idx[-1]]["End_Timestamp"]) - int(rows[idx[0]]["Start_Timestamp"])) / 1e3
    print(f"replay {rep}: wall {wall:8.1f} us, gaps {gap:7.1f} us; per iteration: " + ", ".join(f"{k.split('<')[0].replace('void ','')} {v / 20:5.2f}" for k, v in sorted(dur.items())))
PY
rm -rf $O/t
