#!/bin/bash
# Matrix-core utilisation of the extractor's dense kernels: one counter per run (rocprofv3 --kernel-trace --pmc), summed per
# kernel over the dispatches of `bench.py --config extractor`; the kernel durations come from a plain kernel-trace run.
#   SQ_VALU_MFMA_BUSY_CYCLES: summed over the 1024 SIMDs; one v_mfma_f32_32x32x2_f32 = 64 busy cycles = 4096 flop
#   GRBM_GUI_ACTIVE:          summed over the 8 XCDs -> the clock the kernels actually ran at
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_ext
rm -rf $O; mkdir -p $O
ARGS="--config extractor --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $O/$C -- python3 bench.py $ARGS > $O/$C.json 2> $O/$C.err
  python3 tools/pmc_sum.py $O/$C mlp_ > $O/$C.txt 2>&1
  rm -rf $O/$C
done
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python3 bench.py $ARGS > $O/bench.json 2> $O/stats.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
rm -rf $O/stats
cat $O/*.txt | head -150
