"""Diagnostic build of mlp_chain_wide_kernel (-DCW_CLOCK, REART_LIB=...): average cycles of a wave per weight slab of layers
2 and 3 in each section (store of the slab, barrier, first MFMA group, rest) and per workgroup lifetime."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd import _lib
from reart_amd.networks.feature_extractor import PointNet2Msg2
from reart_amd.synthetic import extractor_state, make_sequence

dev = torch.device("cuda:0")
seq = make_sequence(T=20, n_parts=8, pts_per_part=512, seed=3, with_flow=False)
pts = torch.from_numpy(seq["complete"]).float()
pts = pts - pts.mean(dim=1, keepdim=True)
xyz = (pts / pts.norm(dim=-1).max()).permute(0, 2, 1).contiguous().to(dev)
xyz = torch.cat([xyz, xyz[:18]], 0)
model = PointNet2Msg2(out_dim=64)
model.load_state_dict(extractor_state(model), strict=True)
model = model.to(dev).eval()
L = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    model(xyz); torch.cuda.synchronize()
    L.reart_debug_cw_clock(out, 1)
    model(xyz); torch.cuda.synchronize()
    L.reart_debug_cw_clock(out, 0)
v = np.array(list(out), dtype=np.float64)
n = v[4]
print(f"slabs (per wave) {n:.0f}: store {v[0]/n:.0f}  barrier {v[1]/n:.0f}  first MFMA group {v[2]/n:.0f}  rest of the slab {v[3]/n:.0f} cycles")
m = v[12]
print(f"layer-1 slabs (per wave) {m:.0f}: store {v[8]/m:.0f}  barrier {v[9]/m:.0f}  first MFMA group {v[10]/m:.0f}  rest of the slab {v[11]/m:.0f} cycles")
print(f"workgroup lifetime {v[5]/v[6]:.0f} cycles per wave ({v[6]:.0f} waves)")
