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
eng.step(20); torch.cuda.synchronize()
for rep in range(3):
    eng.step(1); torch.cuda.synchronize()
    lib.reart_debug_phase_clock(buf)
    v = list(buf)
    for w, name, n in ((0, "fwd", 6), (1, "bwd_block", 9)):
        ts = v[16 * w:16 * w + n]
        if w == 1:
            x = v[16:32]
            print("bwd prologue (serialised): hT tile", x[9] - x[0], "G tile", x[10] - x[9], "rt", x[11] - x[10], "W2T", x[12] - x[11], "rest", x[1] - x[12])
        print(name, "deltas (shader cycles; /2400 = us @2.4GHz... s_memtime may tick at 100 MHz):", [ts[i + 1] - ts[i] for i in range(n - 1)], "total", ts[n - 1] - ts[0])
