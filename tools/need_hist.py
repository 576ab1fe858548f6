#!/usr/bin/env python3
"""How many of a wave's 64 queries need each scanned box (stats build):
   REART_LIB=reart_amd/csrc/libreart_hip_stats.so python tools/need_hist.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reart_amd import _lib
eng, seq, model = bench.build_instance(torch.device("cuda:0"), 20, 4096, 10, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
for start in (300, 3000):
    eng.step(start - int(eng.iter.item())); torch.cuda.synchronize()
    lib.reart_debug_prune_stats(buf, 1)
    eng.step(20); torch.cuda.synchronize()
    lib.reart_debug_prune_stats(buf, 1)
    v = list(buf); tot = sum(v[4:8])
    print(f"iteration {start}: scanned boxes {v[2]} (hist total {tot}); needed by 1 lane {v[4]/tot:.2%}, 2-3 lanes {v[5]/tot:.2%}, 4-7 lanes {v[6]/tot:.2%}, 8+ lanes {v[7]/tot:.2%}")
