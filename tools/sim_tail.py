#!/usr/bin/env python3
"""Host-side model of the re-solve's path searches on DUMPED hard solves (tools/exp_tail.py DUMP=...): how deep are the
shortest-path trees of the long searches, and what would bucketed (delta-stepping) rounds buy -- rounds and relaxations
against the one-column-per-step Dijkstra search of lap_jvmw_kernel.  CPU only (numpy); follows the kernel's pipeline in its
plain sequential form: release by row minima, greedy, augmenting row reduction (chains of <= 128 steps), re-pricing of the
unowned columns, then one search per remaining row.
Usage: python tools/sim_tail.py tools/_states/r05_tail_recipe.npz [solve indices ...]"""
import sys
import numpy as np


def cdist32(a, b):
    d = a[:, None, :].astype(np.float32) - b[None, :, :].astype(np.float32)
    s = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return np.sqrt(s.astype(np.float32)).astype(np.float64)


def prepare(C, col, p, chain_cap=128):
    """-> price, owner (col -> row), assigned (row -> col), rows left"""
    n = C.shape[0]
    p = p.copy()
    mx = 1.7320508 * (max(C.max(), 1e-30))
    V = C + p[None, :]
    v1 = V.min(1)
    j1 = V.argmin(1)
    assigned = col.copy()
    owner = -np.ones(n, np.int64)
    for i in range(n):
        j = assigned[i]
        if j >= 0:
            if owner[j] < 0:
                owner[j] = i
            else:
                assigned[i] = -1
    cur = C[np.arange(n), np.maximum(assigned, 0)] + p[np.maximum(assigned, 0)]
    rel = (assigned >= 0) & (cur - v1 > 1e-12 * mx)
    for i in np.nonzero(rel)[0]:
        owner[assigned[i]] = -1
        assigned[i] = -1
    released = int((assigned < 0).sum())
    for i in np.nonzero(assigned < 0)[0]:
        if owner[j1[i]] < 0:
            owner[j1[i]] = i
            assigned[i] = j1[i]
    # re-pricing of the unowned columns
    def tighten():
        own = np.nonzero(owner >= 0)[0]
        urow = C[owner[own], own] + p[own]
        for jh in np.nonzero(owner < 0)[0]:
            m = (C[owner[own], jh] + p[jh] - urow).min()
            if m > 0:
                p[jh] -= m
    tighten()
    free = list(np.nonzero(assigned < 0)[0])
    left = []
    arr = 0
    for i0 in free:
        i = i0
        for step in range(chain_cap + 1):
            v = C[i] + p
            j = int(v.argmin())
            a1 = v[j]
            v[j] = np.inf
            a2 = v.min()
            if step == chain_cap or (a2 == a1 and owner[j] >= 0):
                left.append(i)
                break
            arr += 1
            p[j] += a2 - a1
            k = owner[j]
            owner[j] = i
            assigned[i] = j
            if k < 0:
                break
            assigned[k] = -1
            i = k
    tighten()
    return p, owner, assigned, left, released, arr


def dijkstra(C, p, owner, assigned, i0):
    """one search; -> steps, depth of the tree at the sink, max depth, mu; updates p / owner / assigned in place"""
    n = C.shape[0]
    own = np.nonzero(owner >= 0)[0]
    h = np.full(n, np.nan)
    h[own] = C[owner[own], own] + p[own]
    d = C[i0] + p
    pred = np.full(n, i0)
    depth = np.ones(n, np.int64)
    done = np.zeros(n, bool)
    steps = 0
    order = []
    while True:
        dd = np.where(done, np.inf, d)
        j = int(dd.argmin())
        mu = dd[j]
        steps += 1
        done[j] = True
        if owner[j] < 0:
            sink = j
            break
        order.append(j)
        i = owner[j]
        nd = mu + ((C[i] + p) - h[j])
        better = (~done) & (nd < d)
        d[better] = nd[better]
        pred[better] = i
        depth[better] = depth[j] + 1
    sc = np.array(order, np.int64)
    maxdepth = int(depth[sc].max()) if len(sc) else 1
    p[sc] += mu - d[sc]
    j = sink
    while True:
        i = pred[j]
        jn = assigned[i]
        assigned[i] = j
        owner[j] = i
        if i == i0:
            break
        j = jn
    return steps, int(depth[sink]), maxdepth, mu, d, sc


def delta_stepping(C, p, owner, i0, target=16, w0=None, up=4.0):
    """rounds / relaxations of a bucketed search from i0 on the SAME state (nothing is modified): a bucket = all unsettled
    columns with label < lo + width; inner rounds relax from every bucket member whose label changed, until the bucket is
    stable; width adapts to hold ~target columns.  -> (outer buckets, inner rounds, relaxations, settled below mu)"""
    n = C.shape[0]
    own = np.nonzero(owner >= 0)[0]
    h = np.full(n, np.nan)
    h[own] = C[owner[own], own] + p[own]
    d = C[i0] + p
    settled = np.zeros(n, bool)
    dirty = np.zeros(n, bool)
    width = None
    buckets = rounds = relax = 0
    while True:
        dd = np.where(settled, np.inf, d)
        lo = dd.min()
        if width is None:
            if w0 is None:
                srt = np.sort(dd)
                width = max(srt[min(target, n - 1)] - lo, 1e-300)
            else:
                width = w0
        hi = lo + width
        buckets += 1
        inb = (~settled) & (d < hi)
        dirty[:] = False
        dirty[inb] = True
        while True:
            F = np.nonzero(dirty & (owner >= 0))[0]
            dirty[:] = False
            if len(F) == 0:
                break
            rounds += 1
            relax += len(F)
            for j in F:
                i = owner[j]
                nd = d[j] + ((C[i] + p) - h[j])
                better = (~settled) & (nd < d)
                better[j] = False
                d[better] = nd[better]
                dirty |= better & (nd < hi)
        inb = (~settled) & (d < hi)
        cnt = int(inb.sum())
        sinks = inb & (owner < 0)
        settled |= inb
        if sinks.any():
            mu = d[sinks].min()
            return buckets, rounds, relax, int((settled & (d < mu)).sum())
        if cnt < target // 2:
            width *= up
        elif cnt > 2 * target:
            width *= 0.5


def main():
    z = np.load(sys.argv[1])
    which = [int(a) for a in sys.argv[2:]] or [0, 2, 4]
    tgt = z["tgt"]
    for s in which:
        print(f"== dumped solve {s} (refresh {int(z['idx'][s])}, {z['ms'][s]:.2f} ms on the GPU)")
        for b in range(tgt.shape[0]):
            C = cdist32(z["src"][s, b], tgt[b])
            p, owner, assigned, left, released, arr = prepare(C, z["cols"][s, b].astype(np.int64), z["prices"][s, b])
            tot = 0
            rows = []
            ds_tot = [0, 0, 0]
            for i0 in left:
                bk, rd, rl, st = delta_stepping(C, p, owner, i0)
                steps, dsink, dmax, mu, d, sc = dijkstra(C, p, owner, assigned, i0)
                tot += steps
                rows.append((steps, dsink, dmax, bk, rd, rl))
                ds_tot[0] += bk; ds_tot[1] += rd; ds_tot[2] += rl
            rows.sort(reverse=True)
            print(f"  problem {b}: released {released}, reduction steps {arr}, rows left {len(left)}, search steps {tot}; "
                  f"bucketed: {ds_tot[0]} buckets, {ds_tot[1]} inner rounds, {ds_tot[2]} relaxations; longest searches "
                  f"(steps, path depth, tree depth, buckets, rounds, relaxations): {rows[:4]}")


if __name__ == "__main__":
    main()
