#!/usr/bin/env python3
"""Where a search wave's lifetime goes (stats build: make -C reart_amd/csrc phase):
    REART_LIB=reart_amd/csrc/libreart_hip_phase.so python tools/phase_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reart_amd import _lib

dev = torch.device("cuda:0")
eng, seq, model = bench.build_instance(dev, 20, 4096, 10, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
names = ["prologue", "coarse", "precise", "dense scan", "enqueue", "drain", "barrier wait", "waves"]
done = 0
for target in (300, 1500, 6000):
    eng.step(target - done - 50); done = target
    torch.cuda.synchronize()
    lib.reart_debug_prune_phase(buf, 1)
    eng.step(50)
    torch.cuda.synchronize()
    lib.reart_debug_prune_phase(buf, 1)
    v = list(buf)
    tot = sum(v[:7])
    print(f"iteration {target}: waves/launch {v[7] / 50:.0f}, ticks per wave {tot / max(v[7], 1):.0f}")
    for n, x in zip(names[:7], v[:7]):
        print(f"   {n:14s} {100 * x / tot:5.1f} %   {x / max(v[7], 1):8.0f} ticks/wave")
