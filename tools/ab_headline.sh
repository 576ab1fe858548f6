#!/bin/bash
# A/B of the headline iteration on ONE box: tools/ab_headline.sh tag1 tag2 ...  (tag "base" = the product library; others
# reart_amd/csrc/libreart_hip_<tag>.so), two rounds each, interleaved; prints it/s, search kernel_ms, per-phase eager times.
run() { lib=reart_amd/csrc/libreart_hip_$1.so; [ "$1" = base ] && lib=reart_amd/csrc/libreart_hip.so
  REART_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-secondary --sweep-instances 0 --no-tail --profile-steps 20 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
    print('$1', d['value'], 'it/s  kernel_ms', r.get('kernel_ms'), 'phases_us', {k:round(v*1e3,1) for k,v in d['phases_ms'].items()})
except Exception as e:
    print('$1 no result:', e)
"; }
for rep in 1 2; do for t in "$@"; do run $t; done; done
