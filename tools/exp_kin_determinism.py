#!/usr/bin/env python3
"""Is the kinematic projection loop deterministic run to run -- and if not, WHY?  (missing #3.)
200 iterations of KinematicEngine (README.md:125 configuration) from the reference's kinematic-2 checkpoint on perturbed frames.
  runs A, B   as shipped until round 5 (raced re-solves): bit-identical until one problem returns another assignment.
              At that solve: were the INPUTS bit-equal?  The fp32 cost matrix the solvers see (reart_cdist) is dumped, both
              assignments are summed on it exactly (math.fsum of the fp32 entries), scipy is asked which one IT returns, and
              reart_lap_ties is run on both runs' (assignment, potentials): the rows it flags and the tied rows.
  runs D, E   --deterministic (lap.CANONICAL_TIES): every assignment of every iteration and the parameters compared bit for bit;
              what the tie check costs per solve."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_kinematic_engine_gpu import _model, G, t
from reart_amd.kinematic_engine import KinematicEngine
from reart_amd.utils import lap

ITERS = int(os.environ.get("ITERS", "200"))
dev = torch.device("cuda:0")
k = np.load(os.path.join(G, "kinematic.npz"))
cano = t(k["cano_pc"], dev)
rng = np.random.default_rng(5)
B, N = 9, cano.shape[0]
with torch.no_grad():
    pcs = _model(dev, k, cano)(cano)[0]
pcs = (pcs + t(rng.normal(0, 0.004, (B, N, 3)).astype(np.float32), dev)).contiguous()
pcs = torch.stack([p[torch.from_numpy(rng.permutation(N)).to(dev)] for p in pcs])
comp = torch.cat((pcs[:2], cano[None], pcs[2:]), dim=0)
sel = [torch.from_numpy(rng.permutation(N)[:3000]).to(dev) for f in range(B)]
refs = [comp[f][s] for f, s in enumerate(sel)]
flows = [(comp[f + 1][s] - comp[f][s]) * 0.5 for f, s in enumerate(sel)]


def run(tag, canonical):
    lap.CANONICAL_TIES = canonical
    m = _model(dev, k, cano)
    eng = KinematicEngine(m, cano, pcs, 2, refs, flows, assign_iter=0, assign_gap=1, downsample=2)
    eng.lap_events = []
    out = dict(cols=[], prices=[], src=[], params=[])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(ITERS):
        eng.iteration(i)
        out["cols"].append(eng.lap_state["cols"].clone())
        out["prices"].append(eng.lap_state["prices"].clone())
        out["src"].append(eng._pc_src.clone())
        if (i + 1) % 50 == 0:
            out["params"].append(torch.cat([getattr(m, n_).detach().reshape(-1).clone() for n_ in ("axis_list", "moment_list", "theta_list")]))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in eng.lap_events]
    tb = eng.lap_state.get("tie_breaker")
    print(f"{tag}: canonical={canonical} fallbacks {eng.lap_fallbacks}  {ITERS / el:.1f} it/s  solve ms mean {np.mean(ms[2:]):.3f} p50 {np.median(ms[2:]):.3f}"
          + (f"  ties flagged {tb.flagged} changed {tb.changed} overflows {tb.overflows}" if tb is not None else ""))
    out["tgt"] = eng.tgt_pts
    return out


def compare(a, b, ra, rb):
    print(a, "vs", b, "parameters every 50 iterations:",
          ["equal" if torch.equal(x, y) else f"max diff {float((x - y).abs().max()):.3e}" for x, y in zip(ra["params"], rb["params"])])
    for i, (x, y) in enumerate(zip(ra["cols"], rb["cols"])):
        if not torch.equal(x, y):
            return i
    return None


def dissect(i, ra, rb):
    x, y = ra["cols"][i], rb["cols"][i]
    print(f"  first different assignment at iteration {i}; source points of that solve bit-equal in both runs: {torch.equal(ra['src'][i], rb['src'][i])}"
          f"; assignments of the iteration before equal: {i == 0 or torch.equal(ra['cols'][i - 1], rb['cols'][i - 1])}")
    src, tgt = ra["src"][i], ra["tgt"]
    cost = lap.cdist(src, tgt).cpu().numpy()                       # the fp32 matrix the reference would hand to scipy
    from scipy.optimize import linear_sum_assignment
    from reart_amd import _lib
    for p_ in (x != y).any(1).nonzero().flatten().tolist():
        ca, cb = x[p_].cpu().numpy(), y[p_].cpu().numpy()
        n = len(ca)
        rows = np.arange(n)
        sa, sb = math.fsum(cost[p_][rows, ca].astype(np.float64)), math.fsum(cost[p_][rows, cb].astype(np.float64))
        r_, cs = linear_sum_assignment(cost[p_])
        ss = math.fsum(cost[p_][rows, cs].astype(np.float64))
        diff = np.nonzero(ca != cb)[0]
        print(f"  problem {p_}: rows that differ {diff.tolist()} -> columns {ca[diff].tolist()} vs {cb[diff].tolist()}")
        print(f"    exact sums on the fp32 matrix: run A {sa!r}  run B {sb!r}  scipy {ss!r}   A == B: {sa == sb}  A == scipy: {sa == ss}")
        print(f"    scipy returns run A's assignment: {np.array_equal(cs, ca)}, run B's: {np.array_equal(cs, cb)}")
        sub = cost[p_][np.ix_(diff, np.concatenate((ca[diff], cb[diff])))] if len(diff) <= 4 else None
        if sub is not None:
            print(f"    the fp32 costs of those rows to those columns:\n{np.array2string(sub, precision=10)}")
        for tag, r in (("A", ra), ("B", rb)):
            cols, prices = r["cols"][i], r["prices"][i]
            tb_ = lap.TieBreaker(B, n, dev)
            tb_.launch(src, tgt, cols, prices)
            torch.cuda.synchronize()
            tie, pairs = tb_.tie_host, tb_.pairs_of(p_)
            new, moved = lap.canonical_among_ties(cols[p_].cpu().numpy(), pairs)
            print(f"    reart_lap_ties on run {tag}: flags {tie.tolist()}, tight pairs of problem {p_}: {len(pairs)}; canonical optimum moves {moved} rows"
                  f" -> equals run A's {np.array_equal(new, ca)}, run B's {np.array_equal(new, cb)}")


A, Bq = run("A", False), run("B", False)
i = compare("A", "B", A, Bq)
if i is None:
    print("  A and B never differed in", ITERS, "iterations")
else:
    dissect(i, A, Bq)
D, E = run("D", True), run("E", True)
j = compare("D", "E", D, E)
print("  D vs E (--deterministic): " + ("every assignment of every iteration equal" if j is None else f"first different assignment at iteration {j}"))
if j is not None:
    dissect(j, D, E)
