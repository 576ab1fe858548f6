#!/bin/bash
# A/B of the engine's tuning fields (relax.tuning_from_env) on the headline iteration, ONE box, two rounds.
run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --no-secondary --sweep-instances 0 --no-tail --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'])"; }
for rep in 1 2; do
run X=1
run REART_BWD_PTS=16
run REART_BWD_PTS=64
run REART_FWD_PTS=64
run REART_PRUNE_SPLIT=2
done
