#!/usr/bin/env python3
"""Host-side model of an idea for the HARD re-solves (more than ~24 rows left after the row reduction): augment in PHASES --
one label-setting search from ALL free rows at once (run to the end), then one augmentation per tree that met an unowned column
(the paths of different trees are vertex-disjoint), duals moved by the phase's largest sink label -- against the one-row-at-a-time
searches of lap_jvmw_kernel.  Built on tools/sim_tail.py's sequential restatement of the pipeline; CPU only (numpy).
Result on the slowest dumped projection solve (profiles/r06_sim_phases.txt): exact (same optimum, slack >= -2e-16), but the
phases thin out at once -- 44..92 rows left take 10..32 phases of a WHOLE flood each (12-22 augmentations in the first, 5-11 in
the next two, then one or two per flood), 1.5-2x the settled columns of the sequential searches.  The first phase is what the
forest kernel already gives; phases 2-3 would save about 10 % of a hard solve.  Not built.
Usage: python tools/sim_phases.py tools/_states/r05_tail_proj.npz [solve indices ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from sim_tail import cdist32, prepare, dijkstra

def multi_source_phase(C, p, owner, assigned, free):
    """one phase: forward Dijkstra from ALL free rows at once (label-setting), to the end; the first sink of every tree
    is augmented.  -> augmented count, settled columns, D"""
    n = C.shape[0]
    own = owner >= 0
    h = np.full(n, np.nan)
    oi = np.nonzero(own)[0]
    h[oi] = C[owner[oi], oi] + p[oi]
    free = np.array(free)
    V = C[free] + p[None, :]
    u = V.min(1)
    R = V - u[:, None]
    d = R.min(0)
    root = free[R.argmin(0)]
    pred = root.copy()
    done = np.zeros(n, bool)
    sink_of = {}
    steps = 0
    order = []
    nsinks_seen = 0
    while True:
        dd = np.where(done, np.inf, d)
        j = int(dd.argmin())
        mu = dd[j]
        if not np.isfinite(mu):
            break
        done[j] = True
        steps += 1
        if owner[j] < 0:
            nsinks_seen += 1
            if root[j] not in sink_of:
                sink_of[root[j]] = (j, mu)
                if len(sink_of) == len(free):
                    break
            continue
        order.append(j)
        i = owner[j]
        nd = mu + ((C[i] + p) - h[j])
        better = (~done) & (nd < d)
        d[better] = nd[better]
        pred[better] = i
        root[better] = root[j]
    D = max(v[1] for v in sink_of.values()) if sink_of else 0.0
    sc = np.array([j for j in order if d[j] < D], np.int64)
    # sinks with d < D get their price raised too (they are settled, their tree edge must be tight)
    snk = np.nonzero(done & (owner < 0) & (d < D))[0]
    allsc = np.concatenate([sc, snk])
    p[allsc] += D - d[allsc]
    for r, (j, mu) in sink_of.items():
        while True:
            i = pred[j]
            jn = assigned[i]
            assigned[i] = j
            owner[j] = i
            if i == r:
                break
            j = jn
    return len(sink_of), steps, D, nsinks_seen

def check(C, p, owner, assigned):
    n = C.shape[0]
    a = np.nonzero(assigned >= 0)[0]
    V = C + p[None, :]
    u = V[a, assigned[a]]
    slack = (V[a] - u[:, None]).min()
    return slack

def main():
    z = np.load(sys.argv[1])
    which = [int(a) for a in sys.argv[2:]] or [0]
    tgt = z["tgt"]
    for s in which:
        print(f"== dumped solve {s} ({z['ms'][s]:.2f} ms on the GPU)")
        best = None
        for b in range(tgt.shape[0]):
            C = cdist32(z["src"][s, b], tgt[b])
            t0 = time.time()
            p, owner, assigned, left, released, arr = prepare(C, z["cols"][s, b].astype(np.int64), z["prices"][s, b])
            print(f"  problem {b}: released {released} arr {arr} left {len(left)} ({time.time()-t0:.1f}s)")
            if len(left) < 8:
                continue
            # sequential
            p1, o1, a1 = p.copy(), owner.copy(), assigned.copy()
            tot = 0; lens = []
            for i0 in left:
                st, dsink, dmax, mu, d, sc = dijkstra(C, p1, o1, a1, i0)
                tot += st; lens.append(st)
            cost_seq = C[np.arange(len(a1)), a1].sum()
            print(f"    sequential: {len(left)} searches, {tot} steps; longest {sorted(lens)[-5:]}")
            # phased
            p2, o2, a2 = p.copy(), owner.copy(), assigned.copy()
            free = list(left)
            ph = 0
            while free:
                k, steps, D, ns = multi_source_phase(C, p2, o2, a2, free)
                free = [r for r in free if a2[r] < 0]
                ph += 1
                print(f"    phase {ph}: augmented {k}, settled {steps}, sinks seen {ns}, D {D:.4g}, left {len(free)}, min slack {check(C,p2,o2,a2):.2e}")
                if k == 0:
                    break
            cost_ph = C[np.arange(len(a2)), a2].sum()
            print(f"    cost seq {cost_seq:.9f} phased {cost_ph:.9f}")

main()
