#!/usr/bin/env python3
"""Does the ORDER in which a re-solve takes its free rows change its step count?  Runs the kinematic projection of
bench.py --config kinematic and logs every solve's per-problem sequential steps; run once per library variant
(REART_LIB=...libreart_hip_o1.so = descending order) on ONE box and compare: the assignments are the unique optima, so both
runs see the same problems.   python tools/lap_order_exp.py out.npy
(The variants were builds with a compile-time order switch; since then the order is the racer's index at run time,
reart_lap_resolve_points_race -- REART_RESOLVE_RACERS=1 runs the plain ascending order.)"""
import os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, contextlib
import bench
from reart_amd import run_robot as rr, tail

dev = torch.device("cuda:0")
T, N = 20, 4096
eng, seq, model = bench.build_instance(dev, T, N, 10, seed=2, use_flow=True)
eng.capture(steps_per_graph=50); eng.step(2000); torch.cuda.synchronize()
cano, pcs = eng.caller_clouds()
with torch.no_grad():
    _, seg0, trans0 = model(cano)
seg_s, trans_s, conn_s = tail.extract_structure(seg0, trans0, cano)
result = {"pred_cano_part": seg_s.cpu().numpy(), "pred_pose_list": trans_s.cpu().numpy(), "joint_connection": conn_s.cpu().numpy().tolist(), "cano_idx": 10}
a = rr.build_parser().parse_args(["--model", "kinematic", "--use_flow_loss", "--use_assign_loss", "--assign_iter", "0", "--downsample", "2", "--assign_gap", "1", "--cano_idx", "10"])
with contextlib.redirect_stdout(sys.stderr):
    kin = rr.build_kinematic_from_base(result, cano, pcs, a).to(dev)
t_ = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
loop = rr.make_projection_loop(a, kin, cano, pcs, [t_(r) for r in seq["ref_loc"]], [t_(f) for f in seq["ref_flow"]])
steps, ms = [], []
for it in range(110):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); loop.iteration(it); ev1.record(); torch.cuda.synchronize()
    st = loop.lap_stats
    steps.append(st[:, 2].astype(np.int64) + (st[:, 3].astype(np.int64) >> 8)); ms.append(ev0.elapsed_time(ev1))
steps = np.stack(steps)[10:]          # [100, 19]
np.save(sys.argv[1], steps)
print(os.environ.get("REART_LIB", "default"), "mean ms/iter", np.mean(ms[10:]), "slowest-problem steps (mean over solves)", steps.max(1).mean(), "mean-problem steps", steps.mean())
