#!/bin/bash
# A/B of host-side solver settings on ONE box: VARS="REART_RESOLVE_RACERS=3 REART_RESOLVE_ARR_WGS=16 ..." tools/envvar.sh -- for every
# setting the recipe's assignment phase and 1 500 projection iterations on nao (tools/exp_tail.py), two summary lines each.
out=gpurun_out/r05_var11.txt; rm -f $out
for e in $VARS; do
  echo "=== $e" >> $out
  env $e MODE=recipe timeout 300 python tools/exp_tail.py 2>&1 | grep -v amdgpu.ids | head -2 >> $out
  env $e MODE=projection P_ITERS=1500 timeout 300 python tools/exp_tail.py 2>&1 | grep -v "amdgpu.ids\|joint types" | sed -n 2,3p >> $out
done
cat $out | cut -c1-260
