#!/bin/bash
# usage: TAGS="base x y@morton" tools/ab_tags.sh  -- tools/replay_tail.py on the dumped solves with each library variant
# (tools/mk_variant.sh; "base" = the product library), same box; "@morton": the problems' columns numbered along a Z-order curve first
# (what the product loops do since lap.spatial_order: the dumps hold the solves in the order the loops had when they were taken)
for t in $TAGS; do
  o=""; case $t in *@*) o=${t#*@}; t=${t%@*};; esac
  lib=reart_amd/csrc/libreart_hip_$t.so; [ "$t" = base ] && lib=reart_amd/csrc/libreart_hip.so
  echo "=== $t${o:+ (columns: $o)}"
  ORDER=$o REART_LIB=$lib REPS=${REPS:-3} python tools/replay_tail.py ${DUMPS:-tools/_states/r05s_recipe.npz tools/_states/r05s_proj.npz} 2>&1 | grep "slowest of\|evenly" | cut -c1-165
done
