#!/usr/bin/env python3
"""Where an iteration of the kinematic projection (bench.py --config kinematic) goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reart_amd import run_robot as rr, tail
from reart_amd.networks.pointnet2_utils import index_points
from reart_amd.utils.lap import cdist, linear_sum_assignment_batch, linear_sum_assignment_points

dev = torch.device("cuda:0")
T, N, cano_idx = 20, 4096, 10
eng, seq, model = bench.build_instance(dev, T, N, cano_idx, seed=2)
eng.capture(50); eng.step(2000); torch.cuda.synchronize()
cano, pcs = eng.caller_clouds()
with torch.no_grad():
    _, seg0, trans0 = model(cano)
seg_s, trans_s, conn_s = tail.extract_structure(seg0, trans0, cano)
result = {"pred_cano_part": seg_s.cpu().numpy(), "pred_pose_list": trans_s.cpu().numpy(), "joint_connection": conn_s.cpu().numpy().tolist(), "cano_idx": cano_idx}
a = rr.build_parser().parse_args(["--model", "kinematic", "--use_flow_loss", "--use_assign_loss", "--assign_iter", "0", "--downsample", "2", "--assign_gap", "1", "--cano_idx", str(cano_idx)])
kin = rr.build_kinematic_from_base(result, cano, pcs, a).to(dev)
t_ = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
loop = rr.OperatorLoop(a, kin, cano, pcs, [t_(r) for r in seq["ref_loc"]], [t_(f) for f in seq["ref_flow"]])
WARM = int(os.environ.get("KIN_WARM", "90"))
for i in range(WARM):
    loop.iteration(i)
torch.cuda.synchronize()
import ctypes
from reart_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(lib, "reart_debug_jv_phase"):
    lib.reart_debug_jv_phase((ctypes.c_ulonglong * 320)(), 1)
def T_(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t0)
acc = {}
for i in range(WARM, WARM + 20):
    (out, ms) = T_(lambda: kin(cano)); acc["forward"] = acc.get("forward", 0) + ms
    pc_trans = out[0]
    (pc_src, ms) = T_(lambda: index_points(pc_trans, loop.src_idx)); acc["index"] = acc.get("index", 0) + ms
    (cost, ms) = T_(lambda: cdist(pc_src.detach(), loop.tgt_pts)); acc["cdist"] = acc.get("cdist", 0) + ms
    (res, ms) = T_(lambda: linear_sum_assignment_points(pc_src.detach(), loop.tgt_pts, loop.lap_state, return_stats="full")); acc["lap (points form)"] = acc.get("lap (points form)", 0) + ms
    st = res[2]
    (_, ms) = T_(lambda: loop.iteration(i)); acc["whole iteration (incl. its own lap)"] = acc.get("whole iteration (incl. its own lap)", 0) + ms
    if i % 5 == 0:
        print("lap stats: released", (st[:, 0] & 0xffff).mean(), "left for paths", st[:, 1].mean(), "dijkstra steps", st[:, 2].mean(), "max", st[:, 2].max(), "ARR steps", (st[:, 3] >> 8).mean(), "cert", (st[:, 3] & 255).mean())
for k, v in acc.items():
    print(f"{k:40s} {v / 20:8.3f} ms")

import ctypes
from reart_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(lib, "reart_debug_jv_phase"):
    buf = (ctypes.c_ulonglong * 320)()
    lib.reart_debug_jv_phase(buf, 0)
    v = np.array(list(buf), dtype=np.float64).reshape(32, 10)[:19]
    names = ["search: local + wave arg-min", "search: barrier wait", "search: merge of the waves' minima", "search: row costs + relaxation",
             "search: dual update + path flip", "search: set-up per free row", "before the row reduction", "row reduction", "certificate + outputs", "-"]
    slow = int(v.sum(1).argmax())
    print("s_memtime ticks (100 MHz): share over all workgroups | of the slowest workgroup (%d: %.1f ms over all launches)" % (slow, v[slow].sum() / 1e5))
    for k, n_ in enumerate(names[:9]):
        print(f"   {n_:36s} {100 * v[:, k].sum() / v.sum():5.1f} %   {100 * v[slow, k] / v[slow].sum():5.1f} %")
