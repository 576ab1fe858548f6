#!/bin/bash
# usage: TAGS="base f256 ..." [STEPS=3000] [CONFIG=nao_projection|nao_recipe|kinematic] tools/ab_loop.sh -- the nao projection
# (README.md:125; or the nao recipe README.md:116 / the synthetic kinematic leg, at their own lengths) with each library variant
# (tools/mk_variant.sh; "base" = the product library), --deterministic: tied optima are settled canonically, so every variant walks
# the SAME trajectory (the same problems in the same order, each from the potentials ITS OWN previous solves left) -- what replayed
# dumps (tools/ab_tags.sh) cannot show: the effect of a variant on the state it hands to its next solve.  Same box.
for t in $TAGS; do
  lib=reart_amd/csrc/libreart_hip_$t.so; [ "$t" = base ] && lib=reart_amd/csrc/libreart_hip.so
  if [ "${CONFIG:-nao_projection}" = nao_projection ]; then A="--config nao_projection --steps ${STEPS:-3000} --one-mode"; else A="--config $CONFIG"; fi
  REART_LIB=$lib timeout 400 python bench.py $A --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$t'.ljust(10), 'it/s', d['value'], '| solve ms mean', r['kernel_ms'], 'p50', r['solve_ms_p50'], 'p95', r['solve_ms_p95'], '| ties', d['config'].get('ties'), '| losses', round(d['final_losses']['total Loss'], 9) if isinstance(d.get('final_losses'), dict) else d.get('final_losses', ''), '| whole s', d['config'].get('whole_run_s', ''))"
done
