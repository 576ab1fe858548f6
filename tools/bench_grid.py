import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd import _lib
from reart_amd.synthetic import make_sequence
dev = torch.device("cuda:0")
L = _lib.lib()
seq = make_sequence(T=20, with_flow=False)
pts = torch.from_numpy(seq["complete"]).to(dev)
tgt = pts[1:].contiguous()            # 19 x 4096
def run(name, qry, K, reps=20):
    E, nq = qry.shape[0], qry.shape[1]
    d = torch.empty((E, nq, K), device=dev); i = torch.empty((E, nq, K), dtype=torch.int32, device=dev)
    ws = torch.empty(L.reart_grid_knn_workspace_bytes(E, 4096), dtype=torch.uint8, device=dev)
    def go():
        _lib.check(L.reart_grid_knn(_lib.ptr(tgt), None, E, 4096, _lib.ptr(qry), nq, K, _lib.ptr(d), _lib.ptr(i), _lib.ptr(ws), ws.numel(), _lib.stream()), "g")
    go(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): go()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name:34s} K={K} build+query {dt*1e6:8.1f} us  mean NN dist {d[...,0].sqrt().mean().item():.4f}")
for K in (1, 3):
    run("queries == targets (d=0)", tgt.clone(), K)
    run("queries = previous frame", pts[:-1].contiguous(), K)
    run("queries = targets + N(0,0.005)", (tgt + 0.005 * torch.randn_like(tgt)).contiguous(), K)
    run("queries = targets + N(0,0.05)", (tgt + 0.05 * torch.randn_like(tgt)).contiguous(), K)
    run("queries uniform in box", (torch.rand_like(tgt) * 0.7 - 0.35).contiguous(), K)
