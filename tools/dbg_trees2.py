import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from reart_amd.utils import lap
from reart_amd import _lib
dev = torch.device("cuda:0")
n, B = 1024, 3
L = _lib.lib()
rng = np.random.default_rng(5)
tgt = rng.uniform(-0.3, 0.3, (B, n, 3)).astype(np.float32)
src = (tgt[:, rng.permutation(n)] + rng.normal(0, 0.01, (B, n, 3))).astype(np.float32)
state = {}
s, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
os.environ.pop("REART_LIB", None)
lap.linear_sum_assignment_points(s, t, state)
jump = rng.permutation(n)[: n // 5]
src[:, jump] = rng.uniform(-0.3, 0.3, (B, len(jump), 3)).astype(np.float32)
s = torch.from_numpy(src).to(dev)
col, prices = state["cols"].clone(), state["prices"].clone()
cert = torch.zeros((B,), dtype=torch.int32, device=dev)
racers = 1
ws = _lib.workspace(L.reart_lap_mc_workspace_bytes(B, n, racers), dev)
rc = L.reart_lap_resolve_points_mc(_lib.ptr(s), _lib.ptr(t), B, n, racers, 8, _lib.ptr(col), _lib.ptr(cert), _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), _lib.stream())
torch.cuda.synchronize()
C = oracle.cdist(src, tgt).astype(np.float64)
off0 = L.reart_lap_race_workspace_bytes(B, n, racers)
al = lambda v: (v + 255) // 256 * 256
o = off0
P = ws[o:o + 8 * B * n].view(torch.float64).reshape(B, n).cpu().numpy(); o += al(8 * B * n)
arr = []
for _ in range(6):
    arr.append(ws[o:o + 4 * B * n].view(torch.int32).reshape(B, n).cpu().numpy()); o += al(4 * B * n)
owner, assigned, lst, nxt, tree, tpar = arr
cnt = ws[o:o + 32 * B].view(torch.int32).reshape(B, 8).cpu().numpy()
print("cnt", cnt.tolist())
for b in range(B):
    rows = np.nonzero(assigned[b] >= 0)[0]
    val = C[b] + P[b][None, :]
    u = val[rows, assigned[b][rows]]
    slack = val[rows] - u[:, None]
    print(b, "matched rows", len(rows), "min slack over matched rows x all columns", slack.min(), "consistent owner", all(owner[b][assigned[b][r]] == r for r in rows))
    worst = 0.0
    for j in np.nonzero(tree[b] >= 0)[0]:
        if tpar[b][j] >= 0:
            r = owner[b][j]
            e = C[b][r, tpar[b][j]] + P[b][tpar[b][j]] - (C[b][r, j] + P[b][j])
            worst = max(worst, abs(e))
            if tree[b][tpar[b][j]] != tree[b][j]: print("   parent in another tree!", j, tpar[b][j])
    print("   worst tree-edge slack", worst, "trees", len(np.unique(tree[b][tree[b] >= 0])), "members", int((tree[b] >= 0).sum()), "unowned", int((owner[b] < 0).sum()))
b = 0
shown = 0
for h in np.unique(tree[b][tree[b] >= 0]):
    mem = np.nonzero(tree[b] == h)[0]
    if len(mem) < 3 or shown >= 4: continue
    shown += 1
    print("tree", h, "members", mem.tolist(), "owners", owner[b][mem].tolist(), "tpar", tpar[b][mem].tolist())
    for j in mem:
        r = owner[b][j]
        if r < 0: continue
        u = C[b][r, j] + P[b][j]
        print("   col", j, "row", r, "slack to each member", [(int(t_), float(C[b][r, t_] + P[b][t_] - u)) for t_ in mem if t_ != j])
for b in range(B):
    for j in np.nonzero((tree[b] >= 0) & (tpar[b] >= 0))[0]:
        r = owner[b][j]
        e = C[b][r, tpar[b][j]] + P[b][tpar[b][j]] - (C[b][r, j] + P[b][j])
        if abs(e) > 1e-9:
            h = tree[b][j]; mem = np.nonzero(tree[b] == h)[0]
            print("LOOSE b", b, "col", j, "row", r, "tree", h, "tpar", tpar[b][j], "e", e, "members", mem.tolist(), "tpars", tpar[b][mem].tolist(), "owners", owner[b][mem].tolist(),
                  "slack to members", [float(C[b][r, t_] + P[b][t_] - (C[b][r, j] + P[b][j])) for t_ in mem])
