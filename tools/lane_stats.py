#!/usr/bin/env python3
"""Counters of the per-lane search (lane.hip, stats build):
    make -C reart_amd/csrc stats && REART_LIB=reart_amd/csrc/libreart_hip_stats.so python tools/lane_stats.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reart_amd import _lib
eng, seq, model = bench.build_instance(torch.device("cuda:0"), 20, 4096, 0, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
eng.step(50); torch.cuda.synchronize()
lib.reart_debug_lane_stats(buf, 1)
n = 20
eng.step(n); torch.cuda.synchronize()
lib.reart_debug_lane_stats(buf, 1)
v = list(buf)
w1, w3 = n * 2 * 19 * 64, n * 19 * 64
print("per (wave, cloud): K=1 coarse-passing super boxes %.1f, walk trips %.1f, scan steps %.1f, (query, box) pairs %.1f = %.1f per lane" % (v[0] / w1, v[1] / w1, v[2] / w1, v[3] / w1, v[3] / w1 / 64))
print("                   K=3 coarse-passing super boxes %.1f, walk trips %.1f, scan steps %.1f, (query, box) pairs %.1f = %.1f per lane" % (v[4] / w3, v[5] / w3, v[6] / w3, v[7] / w3, v[7] / w3 / 64))
