#!/usr/bin/env python3
"""Host-side model of a second idea for the hard re-solves: RE-GROW the backward forest (lap_mc_forest_kernel) once its trees are
spent, so that later searches end in a tree again instead of flooding.  Sequential restatement of the pipeline (tools/sim_tail.py)
with the forest as what it is -- a backward Dijkstra search from the unowned columns, 512 rows, prices lowered so that its parent
links are tight -- and searches that end at the first settled column of a live tree.  CPU only (numpy).
Result on the slowest dumped projection solve (profiles/r06_sim_regrow.txt): the trees are never all spent (the searches end in
sinks and trees near THEIR side; roots elsewhere stay unused), so nothing is re-grown -- and a forest of 512 rows saves 6 % of the
settled columns there (32 876 against 35 001 for the problem with 44 rows left): each search still floods 400-750 columns before
it touches the forest.  What would help is a forest that reaches the free rows, i.e. a whole flood per phase: tools/sim_phases.py.
Usage: python tools/sim_regrow.py tools/_states/r05_tail_proj.npz [solve indices ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from sim_tail import cdist32, prepare

def grow_forest(C, p, owner, assigned, rows=512):
    """backward Dijkstra from the unowned columns over alternating paths; -> root[j] (-1 outside), parent[j]; p updated in place"""
    n = C.shape[0]
    sinks = np.nonzero(owner < 0)[0]
    d = np.full(n, np.inf); d[sinks] = 0.0
    root = np.full(n, -1); root[sinks] = sinks
    par = np.full(n, -1)
    inF = np.zeros(n, bool); inF[sinks] = True
    own = np.nonzero(owner >= 0)[0]
    rws = owner[own]
    u = C[rws, own] + p[own]                      # potentials of the owner rows
    lab = np.full(n, np.inf); best = np.full(n, -1)
    # labels of outside owned columns j (row i = owner j): min over forest columns t of red(i,t) + d_t
    for t in sinks:
        v = (C[rws, t] + p[t]) - u + d[t]
        m = v < lab[own]
        lab[own[m]] = v[m]; best[own[m]] = t
    joined = 0; off = 0.0
    while joined < rows:
        cand = np.where(inF, np.inf, lab)
        j = int(cand.argmin())
        if not np.isfinite(cand[j]): break
        off = max(off, cand[j])
        d[j] = max(cand[j], 0.0) if cand[j] > 0 else 0.0
        d[j] = max(d[j], 0.0)
        inF[j] = True; par[j] = best[j]; root[j] = root[best[j]]; joined += 1
        v = (C[rws, j] + p[j]) - u + d[j]
        m = (v < lab[own]) & ~inF[own]
        lab[own[m]] = v[m]; best[own[m]] = j
    F = np.nonzero(inF)[0]
    p[F] -= np.maximum(off - d[F], 0.0)
    return root, par, joined

def search(C, p, owner, assigned, i0, root, live):
    n = C.shape[0]
    own = np.nonzero(owner >= 0)[0]
    h = np.full(n, np.nan); h[own] = C[owner[own], own] + p[own]
    d = C[i0] + p
    pred = np.full(n, i0); done = np.zeros(n, bool)
    order = []; steps = 0
    while True:
        dd = np.where(done, np.inf, d)
        j = int(dd.argmin()); mu = dd[j]; steps += 1; done[j] = True
        if owner[j] < 0 or (root[j] >= 0 and live[root[j]]):
            end = j; break
        order.append(j)
        i = owner[j]
        nd = mu + ((C[i] + p) - h[j])
        better = (~done) & (nd < d)
        d[better] = nd[better]; pred[better] = i
    sc = np.array(order, np.int64)
    if len(sc): p[sc] += mu - d[sc]
    return steps, end, pred

def augment(owner, assigned, i0, end, pred, root, par, live):
    # down the tree first: shift the owners along parent links to the root (a sink)
    r = root[end] if owner[end] >= 0 else end
    if owner[end] >= 0:
        # column end is owned by row a: a moves to par[end], whose owner moves on, ... until the sink
        j = end; carry = owner[end]
        while True:
            k = par[j]
            nxt = owner[k]
            owner[k] = carry; assigned[carry] = k
            if nxt < 0: break
            carry = nxt; j = k
        live[r] = False
    elif root[end] >= 0:
        live[end] = False
    # then the search path to `end`
    j = end
    while True:
        i = pred[j]; jn = assigned[i]
        assigned[i] = j; owner[j] = i
        if i == i0: break
        j = jn

def run(C, p, owner, assigned, left, regrow, rows=512, min_left=6, max_grow=4):
    p, owner, assigned = p.copy(), owner.copy(), assigned.copy()
    n = C.shape[0]
    left = list(left)
    root, par, _ = grow_forest(C, p, owner, assigned, rows)
    live = np.zeros(n, bool); live[np.nonzero(owner < 0)[0]] = True
    grows = 1; tot = 0; per = []
    while left:
        if regrow and grows < max_grow and not live.any() and len(left) >= min_left:
            root, par, _ = grow_forest(C, p, owner, assigned, rows)
            live[:] = False; live[np.nonzero(owner < 0)[0]] = True
            grows += 1
        i0 = left.pop(0)
        st, end, pred = search(C, p, owner, assigned, i0, root, live)
        augment(owner, assigned, i0, end, pred, root, par, live)
        tot += st; per.append(st)
    cost = C[np.arange(n), assigned].sum()
    return tot, grows, cost, per

def main():
    z = np.load(sys.argv[1]); which = [int(a) for a in sys.argv[2:]] or [0]
    tgt = z["tgt"]
    GROW = 190
    for s in which:
        print(f"== dumped solve {s} ({z['ms'][s]:.2f} ms on the GPU)")
        A = B = 0
        for b in range(tgt.shape[0]):
            C = cdist32(z["src"][s, b], tgt[b])
            p, owner, assigned, left, released, arr = prepare(C, z["cols"][s, b].astype(np.int64), z["prices"][s, b])
            if not left: continue
            t0, g0, c0, per0 = run(C, p, owner, assigned, left, False)
            t1, g1, c1, per1 = run(C, p, owner, assigned, left, True)
            assert abs(c0 - c1) < 1e-7, (c0, c1)
            print(f"  problem {b}: left {len(left):3d} | one forest: {t0:6d} steps (+{GROW}) | re-grown x{g1}: {t1:6d} steps (+{GROW*g1}) | cost {c0:.6f}")
            A += t0 + GROW; B += t1 + GROW * g1
        print(f"  slowest-problem view is what the kernel waits for; sums: one forest {A}, re-grown {B} ({100*(B-A)/A:+.1f} %)")
main()
