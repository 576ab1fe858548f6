#!/usr/bin/env python3
"""Wave lifetimes of the work items of one search launch (stats build):
   make -C reart_amd/csrc clock && REART_LIB=reart_amd/csrc/libreart_hip_clock.so python tools/item_clock.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from reart_amd import _lib
eng, seq, model = bench.build_instance(torch.device("cuda:0"), 20, 4096, 10, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 10944   # items of one launch: 3 jobs x 19 frames x 64 query waves x S = 3 slices
buf = (ctypes.c_ulonglong * (3 * n))()
for it in (330, 3000):
    eng.step(it - int(eng.iter.item())); torch.cuda.synchronize()
    lib.reart_debug_item_clock(buf, n)
    v = np.array(list(buf), dtype=np.float64).reshape(n, 3)
    xcc = v[:, 2].astype(int)
    st = np.zeros(n); en = np.zeros(n)
    for x in np.unique(xcc):          # the counters of different XCDs are not aligned: spans per XCD
        m = xcc == x
        t0 = v[m, 0].min(); st[m] = v[m, 0] - t0; en[m] = v[m, 1] - t0
    life = en - st
    print("   per-XCD spans (ticks):", [int(en[xcc == x].max()) for x in np.unique(xcc)], "items per XCD:", [int((xcc == x).sum()) for x in np.unique(xcc)])
    kinds = np.array([1 if (w % 3) < 2 else 3 for w in range(n)])
    print(f"iteration {it}: kernel span {en.max():.0f} ticks; item life mean {life.mean():.0f} p50 {np.percentile(life,50):.0f} p90 {np.percentile(life,90):.0f} p99 {np.percentile(life,99):.0f} max {life.max():.0f}")
    print(f"   K=1 mean {life[kinds==1].mean():.0f}  K=3 mean {life[kinds==3].mean():.0f};  sum of lives / span = {life.sum()/en.max():.0f} waves in flight on average ({life.sum()/en.max()/1024:.2f} per SIMD)")
    print(f"   last start at {st.max():.0f} ({st.max()/en.max():.0%} of the span); items still running at 80% / 90% / 95% of the span: {(en > 0.8*en.max()).sum()} / {(en > 0.9*en.max()).sum()} / {(en > 0.95*en.max()).sum()}")
