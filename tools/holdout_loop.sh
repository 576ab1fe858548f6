#!/bin/bash
# usage: TAGS="base x y" tools/holdout_loop.sh -- the solver's constants at LOOP level on sequences nobody tuned on: the three
# generated hold-out sequences of tools/holdout.sh (recipe's assignment phase + 1 500 iterations of the projection that follows),
# the synthetic kinematic leg and the nao recipe of bench.py, each with every library variant (tools/mk_variant.sh), tied optima
# settled canonically (REART_CANONICAL_TIES=1 / --deterministic) so that every variant solves the SAME problems in the same order,
# each from the potentials its own previous solves left (what replayed dumps cannot show).  Same box.
cd "${GRAFT_REPO_ROOT:-.}"
export REART_CANONICAL_TIES=1
for t in $TAGS; do
  lib=reart_amd/csrc/libreart_hip_$t.so; [ "$t" = base ] && lib=reart_amd/csrc/libreart_hip.so
  export REART_LIB=$lib
  echo "=== $t"
  for set in "h1:11,4,512,10,0.5" "h2:12,8,512,10,2.0" "h3:13,14,256,12,1.0"; do
    SEQ=synthetic:${set#*:} MODE=projection ITERS=6000 ASSIGN_ITER=2000 P_ITERS=1500 timeout 600 python3 tools/exp_tail.py 2>/dev/null | grep "^recipe assignment\|^projection:" | cut -c1-110 | sed "s/^/  ${set%%:*} /"
  done
  timeout 300 python3 bench.py --config kinematic --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  kinematic leg', d['value'], 'it/s, solve ms', d['roofline']['kernel_ms'], 'p50', d['roofline']['solve_ms_p50'])"
  timeout 300 python3 bench.py --config nao_recipe --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  nao recipe', d['value'], 'it/s, solve ms', d['roofline']['kernel_ms'], 'p50', d['roofline']['solve_ms_p50'], 'whole', d['config']['whole_run_s'])"
done
