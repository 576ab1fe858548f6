#!/bin/bash
# A/B of search tuning: prints it/s, kernel_ms, pairs.  Every run under `timeout` (a hang must not take the box).
run() { echo "== $*"; env "$@" timeout 120 python bench.py --no-cpu-baseline --sweep-instances 0 --no-tail --profile-steps 20 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
    print(d['value'], 'it/s  kernel_ms', r.get('kernel_ms'), 'wg_busy_ms', r.get('workgroup_busy_ms'), 'pairs', r.get('executed_pairs_per_launch'), 'phases', {k:round(v*1e3,1) for k,v in d['phases_ms'].items()})
except Exception as e:
    print('no result:', e)
"; }
