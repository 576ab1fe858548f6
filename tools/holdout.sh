#!/bin/bash
# Hold-out for the re-solve's constants: bucket widths, the step threshold, waves per search, chain caps,
# racers -- and round 6's backward growth per row left and the row reduction's tail rule -- were tuned on dumped solves of ONE
# sequence (nao).  Three generated sequences the tuning never saw (other seeds, 4 / 8 / 14 parts, 512 ... 2048 columns, joint
# amplitudes x 0.5 / x 2 / x 1): the README recipe's assignment phase and the projection that follows, their slowest solves
# and an evenly spaced sample dumped (tools/exp_tail.py) and replayed (tools/replay_tail.py) through lap_mw.hip as round 4
# left it (r4), as round 5 left it (r5) and as it is (base), all three linked against the current rest of the library, columns
# along a Z-order curve as the loops number them.  Output: gpurun_out/holdout/holdout.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/holdout; mkdir -p $O tools/_states
: > $O/holdout.txt
for set in "h1:11,4,512,10,0.5" "h2:12,8,512,10,2.0" "h3:13,14,256,12,1.0"; do
  tag=${set%%:*}; spec=${set#*:}
  echo "##### hold-out set $tag: synthetic seed,parts,points per part,frames,amplitude scale = $spec" | tee -a $O/holdout.txt
  SEQ=synthetic:$spec MODE=recipe ITERS=6000 ASSIGN_ITER=2000 KEEP=12 SAMPLE=40 DUMP=tools/_states/ho_${tag}_recipe.npz timeout 600 python3 tools/exp_tail.py 2>&1 | grep -v "amdgpu.ids\|joint types" | head -4 | cut -c1-260 >> $O/holdout.txt
  SEQ=synthetic:$spec MODE=projection ITERS=6000 ASSIGN_ITER=2000 P_ITERS=1500 KEEP=12 SAMPLE=40 DUMP=tools/_states/ho_${tag}_proj.npz timeout 900 python3 tools/exp_tail.py 2>&1 | grep -v "amdgpu.ids\|joint types" | grep "projection\|solves " | cut -c1-260 >> $O/holdout.txt
  for lib in ${LIBS:-r4 r5 base}; do
    so=reart_amd/csrc/libreart_hip_$lib.so; [ $lib = base ] && so=reart_amd/csrc/libreart_hip.so
    echo "=== $lib" >> $O/holdout.txt
    for d in recipe proj; do
      [ -f tools/_states/ho_${tag}_$d.npz ] && KEEP=12 ORDER=morton REART_LIB=$so REPS=3 timeout 600 python3 tools/replay_tail.py tools/_states/ho_${tag}_$d.npz 2>&1 | grep "solves of\|slowest of\|evenly" | cut -c1-200 >> $O/holdout.txt
    done
  done
done
cat $O/holdout.txt
