#!/usr/bin/env python3
"""Where the per-wave re-solve (csrc/lap_mw.hip) spends its time on the base recipe's refreshes (nao, 9 x 1024^2): device
clocks of the diagnostic build (make -C reart_amd/csrc phase; REART_LIB=.../libreart_hip_phase.so).
Usage: gpurun -- 'make -C reart_amd/csrc phase && REART_LIB=$PWD/reart_amd/csrc/libreart_hip_phase.so python tools/exp_mw.py'"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd import _lib
from reart_amd.networks.model import BaseModel
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.relax import RelaxEngine
from reart_amd.utils import lap

dev = torch.device("cuda:0")
from reart_amd.data import load_nao_demo
g = load_nao_demo()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cano, pcs, cano_idx = t(g["cano"]), t(g["pc_list"]), int(g["cano_idx"])
gt_pos = t(g["complete_gt_pc_list"])
rng = np.random.default_rng(0)
sel = [torch.from_numpy(rng.permutation(gt_pos.shape[1])[:3000]).to(dev) for _ in range(pcs.shape[0])]
refs = [gt_pos[k][s] for k, s in enumerate(sel)]
flows = [t(g["gt_flow_list"][k])[s] for k, s in enumerate(sel)]
torch.manual_seed(2)
model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
eng = RelaxEngine(cano, pcs, model, cano_idx, refs, flows, n_iter=15000, seed=2)
i = eng.capture(steps_per_graph=10)
eng.step(int(os.environ.get("START", 8000)) - i)
B, N = pcs.shape[:2]; nf = N // 4
zero = torch.zeros(1, dtype=torch.long, device=dev)
src = farthest_point_sample(cano[None], nf, start=zero, cuda_mode=True)
tgt = farthest_point_sample(pcs, nf, start=zero.expand(B), cuda_mode=True)
tgt_pts = index_points(pcs, tgt).contiguous()
L = _lib.lib()
has_phase = hasattr(L, "reart_debug_mw_phase")
buf = (ctypes.c_ulonglong * 640)()
state = {}
names = ["setup", "arr wall", "sap wall", "arr busy/8", "sap busy/8", "lock wait/8", "wasted steps", "all steps", "searches", "longest"]
for k in range(int(os.environ.get("REPS", 8))):
    eng.peek_forward()
    src_pts = index_points(eng.pc_trans, src.expand(B, nf)).contiguous()
    if k == 0:
        out = lap.linear_sum_assignment_points(src_pts, tgt_pts, state)
    else:
        forms = [(1, ("mc", 8))]
        for racers, form in forms:
            lap.RESOLVE_RACERS = racers
            js = {"prices": state["prices"].clone(), "cols": state["cols"].clone()}
            if has_phase:
                L.reart_debug_mw_phase(buf, 1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out, fb, st = lap.linear_sum_assignment_points(src_pts, tgt_pts, js, return_stats="full", per_wave=form)
            torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
            line = f"refresh {k} racers {racers:2d} form {form}: {ms:6.2f} ms fb {fb} left {st[:, 1].tolist()} steps {st[:, 2].tolist()} arr {(st[:, 3] >> 8).tolist()} conf {js.get('commit_conflicts', np.zeros(1)).tolist()}"
            print(line)
            if hasattr(L, "reart_debug_mw_step"):
                sb = (ctypes.c_ulonglong * 8)()
                L.reart_debug_mw_step(sb, 1)
                tot = sum(sb[:6]) or 1
                print("    step sections (s_memtime ticks, problem 0 wave 0; share): " + " ".join(f"{nm} {sb[q]} ({100 * sb[q] / tot:.0f}%)" for q, nm in enumerate(["argmin", "barrier", "merge", "reads", "relax", "between"])) + f" | steps {st[0, 2]} -> {tot / max(st[0, 2], 1):.0f} ticks per step")
            if has_phase:
                L.reart_debug_mw_phase(buf, 0)
                a = np.array(buf[:], dtype=np.float64).reshape(64, 10)[:B]
                a[:, :6] /= 100.0        # 100 MHz ticks -> us
                a[:, 3:6] /= 8.0
                w = int(np.argmax(a[:, 1] + a[:, 2]))
                print("    slowest problem", w, " ".join(f"{n_} {a[w, q]:.0f}" for q, n_ in enumerate(names)))
                print("    mean           ", " ".join(f"{n_} {a[:, q].mean():.0f}" for q, n_ in enumerate(names)))
        if hasattr(L, "reart_debug_jv_hist"):      # the one-search-at-a-time solver on the same problem: how long are its searches?
            hb = (ctypes.c_ulonglong * 32)()
            L.reart_debug_jv_hist(hb, 1)
            lap.RESOLVE_RACERS = 1
            jo = {"prices": state["prices"].clone(), "cols": state["cols"].clone()}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, fb, st = lap.linear_sum_assignment_points(src_pts, tgt_pts, jo, return_stats="full", per_wave=False)
            torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
            L.reart_debug_jv_hist(hb, 0)
            h = np.array(hb[:], dtype=np.int64).reshape(16, 2)
            print(f"    sequential: {ms:.2f} ms left {st[:, 1].tolist()} steps {st[:, 2].tolist()}; searches by length (<=2^b: count/steps): "
                  + " ".join(f"{1 << b_}:{h[b_, 0]}/{h[b_, 1]}" for b_ in range(16) if h[b_, 0]))
        state = js
    cols = torch.from_numpy(np.stack([c for _, c in out])).to(dev)
    eng.set_assignment(src[0], tgt.gather(1, cols), 0.3)
    eng.step(5)
