#!/bin/bash
# usage: tools/ab_replay.sh out.txt lib_tag ...   (tag "base" = the product library): tools/replay_tail.py on the dumped solves with each
# library, twice in alternation (same box, same problems)
out=$1; shift
for rep in 1 2; do
  for t in "$@"; do
    lib=reart_amd/csrc/libreart_hip_$t.so; [ "$t" = base ] && lib=reart_amd/csrc/libreart_hip.so
    echo "=== $t (pass $rep)" >> $out
    REART_LIB=$lib REPS=${REPS:-3} timeout 600 python tools/replay_tail.py ${DUMPS:-tools/_states/r05s_recipe.npz tools/_states/r05s_proj.npz} 2>&1 | grep -v "amdgpu.ids\|per solve" >> $out
  done
done
