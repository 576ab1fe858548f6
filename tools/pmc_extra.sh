#!/bin/bash
# Extra counter passes for the search kernel (which pipeline is busy): one counter per run, --kernel-trace only.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_extra
rm -rf $O; mkdir -p $O
for C in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $O/$C -- python3 bench.py --steps 100 --warmup 100 --no-graph --profile-steps 0 --sweep-instances 0 --no-tail --no-cpu-baseline --no-secondary > /dev/null 2> $O/$C.err
  python3 tools/pmc_sum.py $O/$C knn_group | tail -1
  rm -rf $O/$C
done
