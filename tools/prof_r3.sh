#!/bin/bash
# round 3: counters + phase clocks of the search kernel at HEAD (one gpurun call)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof3; rm -rf $O; mkdir -p $O
python -m pytest tests/test_sweep_gpu.py -m gpu -q -x -k sweep_command 2>&1 | tail -30 > $O/sweep_test.txt
REART_LIB=reart_amd/csrc/libreart_hip_phase.so timeout 300 python tools/phase_prof.py > $O/phase.txt 2>&1
CLEAN="--sweep-instances 0 --no-tail --no-cpu-baseline --no-secondary"
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $O/pmc_$C -- python3 bench.py --steps 300 --warmup 150 --no-graph --profile-steps 0 $CLEAN > $O/pmc_$C.json 2> $O/pmc_$C.err
  python3 tools/pmc_sum.py $O/pmc_$C knn_group > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
tail -3 $O/pmc_*.txt; cat $O/phase.txt; cat $O/sweep_test.txt
