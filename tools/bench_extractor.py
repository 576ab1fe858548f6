#!/usr/bin/env python3
"""One-time pre-loop cost: PointNet2Msg2 on 2*(T-1) = 38 clouds of 4096 points (flow_utils.py:123-124)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd.networks.feature_extractor import PointNet2Msg2
from reart_amd.synthetic import make_sequence
from reart_amd.synthetic import extractor_state

dev = torch.device("cuda:0")
seq = make_sequence(T=20, with_flow=False)
pts = torch.from_numpy(seq["complete"]).to(dev)
pts = (pts - pts.mean(dim=1, keepdim=True)); pts = pts / pts.norm(dim=-1).max()
xyz = torch.cat([pts[:-1], pts[1:]]).permute(0, 2, 1).contiguous()  # [38,3,4096]
model = PointNet2Msg2(64); model.load_state_dict(extractor_state(model)); model = model.to(dev).eval()
B = xyz.shape[0]
s = (torch.zeros(B, dtype=torch.long, device=dev), torch.zeros(B, dtype=torch.long, device=dev))
for _ in range(2): f = model(xyz, fps_start=s)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): f = model(xyz, fps_start=s)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
flop = 9.55e9 * B
print(f"extractor: {B} clouds x 4096 pts: {dt*1e3:.2f} ms  ({flop/dt/1e12:.1f} TFLOP/s dense-equivalent, {dt/B*1e3:.3f} ms/cloud); feat {tuple(f.shape)} finite={bool(torch.isfinite(f).all())}")
