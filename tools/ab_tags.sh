#!/bin/bash
# usage: TAGS="base x y" tools/ab_tags.sh  -- tools/replay_tail.py on the dumped solves with each library variant (tools/mk_variant.sh), same box
for t in $TAGS; do
  lib=reart_amd/csrc/libreart_hip_$t.so; [ "$t" = base ] && lib=reart_amd/csrc/libreart_hip.so
  echo "=== $t"
  REART_LIB=$lib REPS=${REPS:-3} python tools/replay_tail.py ${DUMPS:-tools/_states/r05s_recipe.npz tools/_states/r05s_proj.npz} 2>&1 | grep "slowest of\|evenly" | cut -c1-165
done
