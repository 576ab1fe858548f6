#!/usr/bin/env python3
"""Shader-clock stamps inside base_fwd_kernel / base_bwd_block_kernel (workgroup 1, thread 0).
    make -C reart_amd/csrc stats && REART_LIB=reart_amd/csrc/libreart_hip_stats.so python tools/phase_clock.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reart_amd import _lib
eng, seq, model = bench.build_instance(torch.device("cuda:0"), 20, 4096, 0, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
pbuf = (ctypes.c_ulonglong * 8)()
TICK_US = 1e-2        # replaced below by the measured rate of s_memtime (against the 100 MHz wall clock stamped next to it)
eng.step(20); torch.cuda.synchronize()
for rep in range(3):
    eng.step(1); torch.cuda.synchronize()
    lib.reart_debug_phase_clock(buf)
    v = list(buf)
    for w, name, n in ((0, "fwd", 6), (1, "bwd_block", 9)):
        ts = v[16 * w:16 * w + n]
        wall = (v[16 * w + 13] - v[16 * w + 12]) * 1e-2           # us between the first and the latest stamp (100 MHz wall clock)
        if wall > 0:
            TICK_US = wall / (ts[n - 1] - ts[0])
            print(f"{name}: s_memtime runs at {1e-3 / TICK_US:.3f} GHz here ({ts[n - 1] - ts[0]} ticks in {wall:.2f} us)")
        if w == 1:
            x = v[16:32]
            print("bwd prologue (serialised): hT tile", x[9] - x[0], "G tile", x[10] - x[9], "rt", x[11] - x[10], "W2T", x[12] - x[11], "rest", x[1] - x[12])
        print(name, "deltas (s_memtime ticks):", [ts[i + 1] - ts[i] for i in range(n - 1)], "total", ts[n - 1] - ts[0],
              f"= {(ts[n - 1] - ts[0]) * TICK_US:.2f} us: the lifetime of ONE workgroup (block 1)")
    fz = v[8:16]          # [0][8] body start, [9] body end, [10] after the ticket, [14] kernel entry (before the bookkeeping's speculative part)
    print(f"finalize, workgroup 1: main work (column sums, Gram-Schmidt backward, Adam; the bookkeeping's speculative part -- loss partials, pow / "
          f"sqrt / temperature -- runs on a fifth wave beside it) {(fz[1] - fz[6]) * TICK_US:.2f} us | barrier + ticket round trip "
          f"{(fz[2] - fz[1]) * TICK_US:.2f} us = lifetime {(fz[2] - fz[6]) * TICK_US:.2f} us")
    if hasattr(lib, "reart_debug_post_clock"):
        lib.reart_debug_post_clock(pbuf)
        p = list(pbuf)
        print("post_kernel, one workgroup of each kind (lifetime, us): flow blend %.2f | Chamfer gradient %.2f | launch order %.2f | profile %.2f; "
              "(kernel durations by rocprofv3: profiles/*kernel_stats_clean.csv)" % tuple([(p[2 * k + 1] - p[2 * k]) * TICK_US for k in range(4)]))
