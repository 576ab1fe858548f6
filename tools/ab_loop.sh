#!/bin/bash
# usage: TAGS="base f256 ..." [STEPS=3000] tools/ab_loop.sh -- the nao projection (README.md:125) with each library variant
# (tools/mk_variant.sh; "base" = the product library), --deterministic: tied optima are settled canonically, so every variant walks
# the SAME trajectory (the same problems in the same order, each from the potentials ITS OWN previous solves left) -- what replayed
# dumps (tools/ab_tags.sh) cannot show: the effect of a variant on the state it hands to its next solve.  Same box.
for t in $TAGS; do
  lib=reart_amd/csrc/libreart_hip_$t.so; [ "$t" = base ] && lib=reart_amd/csrc/libreart_hip.so
  REART_LIB=$lib timeout 400 python bench.py --config nao_projection --steps ${STEPS:-3000} --no-cpu-baseline --one-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$t'.ljust(10), 'it/s', d['value'], '| solve ms mean', r['kernel_ms'], 'p50', r['solve_ms_p50'], 'p95', r['solve_ms_p95'], '| ties', d['config'].get('ties'), '| losses', round(d['final_losses']['total Loss'], 9) if 'final_losses' in d else '')"
done
