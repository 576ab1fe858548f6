#!/usr/bin/env python3
"""Counters of quad.hip (stats build):  REART_SEARCH=quad REART_LIB=reart_amd/csrc/libreart_hip_stats.so python tools/quad_stats.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reart_amd import _lib
eng, seq, model = bench.build_instance(torch.device("cuda:0"), 20, 4096, 10, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
eng.step(300); torch.cuda.synchronize()
lib.reart_debug_quad_stats(buf, 1)
eng.step(20); torch.cuda.synchronize()
lib.reart_debug_quad_stats(buf, 1)
v = list(buf)
for k, name in ((0, "K=1"), (4, "K=3")):
    n = max(v[k], 1)
    print(f"{name}: per (16-query wave, 128-box pass): coarse-passing boxes {v[k+1]/n:.1f}, test steps {v[k+2]/n:.1f}, scan steps {v[k+3]/n:.1f}   (passes {v[k]})")
