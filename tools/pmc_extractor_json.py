#!/usr/bin/env python3
"""profiles/rNN_pmc_extractor_mfma.json from the outputs of tools/pmc_extractor.sh:
    python tools/pmc_extractor_json.py gpurun_out/pmc_ext > profiles/r03_pmc_extractor_mfma.json
Per dense kernel of the extractor: matrix-core busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over
GRBM_GUI_ACTIVE / 8 XCDs (both summed over the same dispatches of `bench.py --config extractor --steps 10 --warmup 2`)."""
import csv, json, os, re, sys

d = sys.argv[1]


def read(counter):
    out, name = {}, None
    for line in open(os.path.join(d, counter + ".txt")):
        if not line.startswith(" "):
            name = line.strip().replace("void ", "")
        else:
            m = re.match(r"\s+(\S+)\s+(\d+)\s+/dispatch\s+([\d.]+)\s+\(n=(\d+)\)", line)
            if m:
                out[name] = (float(m.group(2)), int(m.group(4)))
    return out


mfma, gui = read("SQ_VALU_MFMA_BUSY_CYCLES"), read("GRBM_GUI_ACTIVE")
dur = {}
for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
    dur[r["Name"].replace("void ", "").split("(")[0]] = (float(r["TotalDurationNs"]), int(r["Calls"]))
kernels, tot_m, tot_g = {}, 0.0, 0.0
for k in sorted(mfma):
    m, n = mfma[k]
    g, _ = gui[k]
    busy = (m / 1024.0) / (g / 8.0)
    t = dur.get(k, (0.0, 0))
    kernels[k] = {"dispatches": n, "mfma_busy_cycles_per_simd": round(m / 1024.0 / n, 1), "gpu_cycles_per_xcd": round(g / 8.0 / n, 1),
                  "mfma_busy_frac": round(busy, 4), "issued_gflop_per_dispatch": round(m / 64.0 * 4096 / n / 1e9, 3),
                  "avg_us_plain_trace": round(t[0] / max(t[1], 1) / 1e3, 2),
                  "clock_ghz": round(g / 8.0 / n / (t[0] / max(t[1], 1)), 3) if t[1] else None}
    tot_m += m / 1024.0
    tot_g += g / 8.0
bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
print(json.dumps({
    "what": "matrix-core utilisation of the correspondence extractor's dense kernels (v_mfma_f32_32x32x2_f32), 38 clouds x 4096 points",
    "command": "tools/pmc_extractor.sh: rocprofv3 --kernel-trace --pmc <one counter per run> -- python3 bench.py --config extractor --steps 10 "
               "--warmup 2 --no-cpu-baseline --no-secondary; durations from a separate --kernel-trace --stats run",
    "notes": ["SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; one 32x32x2 f32 MFMA = 64 busy cycles = 4096 flop",
              "GRBM_GUI_ACTIVE is summed over the 8 XCDs",
              "mlp_gemm_kernel<2, 16> = the ten small layers left outside the fused chains (sa3, fp3, fp2, fp1, conv1)"],
    "kernels": kernels,
    "all_dense_kernels_mfma_busy_frac": round(tot_m / tot_g, 4),
    "forward_ms_under_rocprof_stats": bench.get("ms_per_step"),
}, indent=1))
