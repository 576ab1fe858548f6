#!/usr/bin/env python3
"""The counter summary of the search launch that bench.py's `roofline.traffic` cites, from the per-counter sums
tools/collect_profiles.sh leaves (pmc_<COUNTER>.txt, written by tools/pmc_sum.py), STAMPED with the SHA-256 of the kernel
sources it was measured on: bench.py compares the stamps with the sources it runs and says when they differ, so a stale
number cannot pass for a measurement of the current kernel.   python tools/pmc_search_json.py gpurun_out/prof4 > profiles/r04_pmc_search.json"""
import hashlib, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
SOURCES = ("reart_amd/csrc/prune.hip", "reart_amd/csrc/step.hip", "reart_amd/csrc/model.hip", "reart_amd/csrc/internal.h")


def stamp():
    return {p: hashlib.sha256(open(os.path.join(root, p), "rb").read()).hexdigest()[:16] for p in SOURCES}


per, n = {}, 0
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    f = os.path.join(d, f"pmc_{c}.txt")
    if not os.path.exists(f):
        continue
    m = re.search(rf"{c}\s+\d+\s+/dispatch\s+([0-9.]+)\s+\(n=(\d+)\)", open(f).read())
    if m:
        per[c], n = float(m.group(1)), int(m.group(2))
out = {"kernel": "knn_group_kernel<false> (Chamfer K=1 both directions + flow K=3, one launch)",
       "command": "rocprofv3 --kernel-trace --pmc <C> -f csv -- python3 bench.py --steps 300 --warmup 150 --no-graph --profile-steps 0 "
                  "--sweep-instances 0 --no-tail --no-cpu-baseline --no-secondary (one pass per counter, tools/collect_profiles.sh)",
       "dispatches_averaged": n, "per_launch": per, "measured_on_sources_sha256_16": stamp(),
       "slow_box": os.path.exists(os.path.join(d, "slowbox.txt"))}
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    # FETCH_SIZE / WRITE_SIZE count the L2's memory-side requests in KiB (MI355X_MICROARCH.md, HBM section): fabric traffic,
    # Infinity-Cache hits included -- an upper bound of HBM traffic; reported raw as in rounds 2-3
    out["hbm_bytes_per_launch"] = int(round((per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024))
print(json.dumps(out, indent=1))
