import sys, json
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], "it/s", d["ms_per_step"], "ms", {k: round(v * 1e3, 1) for k, v in d["phases_ms"].items()})
