#!/bin/bash
# Copies the summaries of a collection (tools/collect_profiles.sh -> gpurun_out/prof6/) into profiles/ under their round-6 names.
# The text files that carry hand-written headers (r06_run_nao_projection.txt) are rebuilt from their header + the new runs;
# profiles/r06_exp_kin_determinism.txt (the proof of the tie at the first diverging solve) is only replaced when the new run
# actually diverged (otherwise the new output goes to r06_exp_kin_determinism_second_box.txt).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof6; P=profiles
for f in bench_clean_under_rocprof bench_default bench_driver_args bench_driver_default bench_kinematic bench_kinematic_under_rocprof bench_nao \
         bench_nao_projection_1500_under_rocprof bench_nao_recipe bench_nao_recipe_under_rocprof; do cp $O/$f.json $P/r06_$f.json; done
cp $O/kernel_stats_clean.csv $P/r06_kernel_stats_clean.csv
for k in kinematic nao_projection nao_recipe; do cp $O/kernel_stats_${k}_window.csv $P/r06_kernel_stats_$k.csv; done
cp $O/pmc_search.json $P/r06_pmc_search.json
cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_SQ_INSTS_VALU.txt $O/pmc_SQ_ACTIVE_INST_VALU.txt $O/pmc_GRBM_GUI_ACTIVE.txt $O/pmc_SQ_WAVES.txt \
    $O/pmc_SQ_INSTS_SALU.txt $O/pmc_SQ_INSTS_LDS.txt > $P/r06_pmc_search_raw.txt
for f in exp_tail_projection exp_tail_recipe solve_spans_nao_projection iteration_glue_nao_projection solve_gaps_nao_projection solve_gaps_nao_recipe \
         replay_kernels_proj_after replay_kernels_recipe_after; do [ -s $O/$f.txt ] && cp $O/$f.txt $P/r06_$f.txt; done
cp $O/sweep_recipe_20x15000_energy.json $P/r06_sweep_recipe_20x15000_energy.json
python3 - <<'PY'
O, P = "gpurun_out/prof6", "profiles"
head = open(f"{P}/r06_run_nao_projection.txt").read().split("--- run 1")[0]
body = ""
for k, (f, lab) in enumerate((("run_nao_projection_det_1.txt", "--deterministic, default"), ("run_nao_projection_det_2.txt", "--deterministic, default"),
                              ("run_nao_projection_nodet.txt", "--no-deterministic")), 1):
    body += f"--- run {k} ({lab})\n" + open(f"{O}/{f}").read().rstrip("\n") + "\n"
open(f"{P}/r06_run_nao_projection.txt", "w").write(head + body)
kd = open(f"{O}/exp_kin_determinism.txt").read()
if "never differed" in kd:
    note = ("# a later collection of tools/exp_kin_determinism.py (another box): this time the two raced runs A and B met no tie that fell\n"
            "# differently within 200 iterations, so there is no diverging solve to take apart -- the proof is in r06_exp_kin_determinism.txt;\n"
            "# the deterministic pair D / E agrees in every assignment, as there.\n")
    open(f"{P}/r06_exp_kin_determinism_second_box.txt", "w").write(note + kd)
else:
    open(f"{P}/r06_exp_kin_determinism.txt", "w").write(kd)
PY
cat $O/probe.txt; ls $O/slowbox.txt 2>/dev/null && echo "NOTE: a slow-search box"
