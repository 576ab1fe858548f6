#!/bin/bash
# usage: run_variants.sh out_prefix tag1 tag2 ...   (tag "base" = the product library); MODES="recipe projection" P_ITERS=1500
out=$1; shift
for t in "$@"; do
  lib=reart_amd/csrc/libreart_hip_$t.so; [ "$t" = base ] && lib=reart_amd/csrc/libreart_hip.so
  echo "=== $t" >> $out.txt
  for m in ${MODES:-recipe projection}; do
    if [ $m = recipe ]; then REART_LIB=$lib MODE=recipe timeout 300 python tools/exp_tail.py 2>&1 | grep -v amdgpu.ids | head -${HEAD:-6} >> $out.txt
    else REART_LIB=$lib MODE=projection P_ITERS=${P_ITERS:-1500} timeout 300 python tools/exp_tail.py 2>&1 | grep -v "amdgpu.ids\|joint types" | sed -n 2,${HEAD2:-7}p >> $out.txt; fi
  done
done
