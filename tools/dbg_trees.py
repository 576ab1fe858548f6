import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from reart_amd.utils import lap
dev = torch.device("cuda:0")
n = int(os.environ.get("N", 1024)); B = 3
rng = np.random.default_rng(n)
tgt = rng.uniform(-0.3, 0.3, (B, n, 3)).astype(np.float32)
src = (tgt[:, rng.permutation(n)] + rng.normal(0, 0.01, (B, n, 3))).astype(np.float32)
state = {}
lap.RESOLVE_RACERS = int(os.environ.get("RACERS", 1))
for k in range(5):
    if k == 2:
        jump = rng.permutation(n)[: n // 5]
        src[:, jump] = rng.uniform(-0.3, 0.3, (B, len(jump), 3)).astype(np.float32)
    elif k != 4:
        src = (src + rng.normal(0, 0.002, src.shape)).astype(np.float32)
    s, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    out, fb, st = lap.linear_sum_assignment_points(s, t, state, return_stats="full")
    ref = oracle.linear_sum_assignment(oracle.cdist(src, tgt))
    C = oracle.cdist(src, tgt)
    mism = [int((c != ref[b][1]).sum()) for b, (r, c) in enumerate(out)]
    cost = [float(C[b][np.arange(n), c].sum() - C[b][np.arange(n), ref[b][1]].sum()) for b, (r, c) in enumerate(out)]
    print(k, "fb", fb, "mismatch", mism, "cost diff", cost, "stats", st.tolist(), flush=True)

# direct call on the last problem's predecessor state: inspect the published duals
from reart_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(5)
tgt = rng.uniform(-0.3, 0.3, (B, n, 3)).astype(np.float32)
src = (tgt[:, rng.permutation(n)] + rng.normal(0, 0.01, (B, n, 3))).astype(np.float32)
state = {}
s, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
lap.linear_sum_assignment_points(s, t, state)
jump = rng.permutation(n)[: n // 5]
src[:, jump] = rng.uniform(-0.3, 0.3, (B, len(jump), 3)).astype(np.float32)
s = torch.from_numpy(src).to(dev)
col, prices = state["cols"].clone(), state["prices"].clone()
p_before = prices.cpu().numpy().copy(); c_before = col.cpu().numpy().copy()
cert = torch.zeros((B,), dtype=torch.int32, device=dev)
racers = 1
ws = _lib.workspace(L.reart_lap_mc_workspace_bytes(B, n, racers), dev)
rc = L.reart_lap_resolve_points_mc(_lib.ptr(s), _lib.ptr(t), B, n, racers, 8, _lib.ptr(col), _lib.ptr(cert), _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), _lib.stream())
torch.cuda.synchronize()
print("rc", rc, "cert", cert.tolist())
C = oracle.cdist(src, tgt).astype(np.float64)
p = prices.cpu().numpy(); c4 = col.cpu().numpy()
for b in range(B):
    val = C[b] + p[b][None, :]
    own = val[np.arange(n), c4[b]]
    viol = own - val.min(1)
    bad = np.nonzero(viol > 1e-13)[0]
    print(b, "perm ok", len(set(c4[b].tolist())) == n, "max violation", viol.max(), "rows violating", len(bad), bad[:10].tolist(), viol[bad[:10]].tolist())

    am = val.argmin(1)
    cols_, cnts = np.unique(am[bad], return_counts=True)
    top = cols_[np.argsort(-cnts)][:6]
    # workspace layout: race bytes | price | owner | assigned | list | next | tree | tpar | cnt
    off0 = L.reart_lap_race_workspace_bytes(B, n, racers)
    al = lambda v: (v + 255) // 256 * 256
    o_tree = off0 + al(8 * B * n) + 4 * al(4 * B * n)
    tree = ws[o_tree:o_tree + 4 * B * n].view(torch.int32).reshape(B, n).cpu().numpy()
    tpar = ws[o_tree + al(4 * B * n):o_tree + al(4 * B * n) + 4 * B * n].view(torch.int32).reshape(B, n).cpu().numpy()
    print("   attractive columns", top.tolist(), "price before", p_before[b][top].tolist(), "after", p[b][top].tolist(), "tree", tree[b][top].tolist(), "tpar", tpar[b][top].tolist(),
          "owner row now", [int(np.nonzero(c4[b] == j)[0][0]) for j in top], "owner before", [int(np.nonzero(c_before[b] == j)[0][0]) if (c_before[b] == j).any() else -1 for j in top])
    print("   tree sizes", np.bincount(tree[b][tree[b] >= 0]).tolist())
