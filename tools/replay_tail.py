#!/usr/bin/env python3
"""Replay DUMPED assignment re-solves (tools/exp_tail.py DUMP=... [SAMPLE=n]) through the product entry point
(reart_lap_resolve_points_mc via lap.InPlaceResolve), every solve from the state it started from in the run: the SAME problems
for every library variant (REART_LIB=...), which two runs of a loop never give (the trajectories are chaotic).
Prints per group (the run's slowest solves / the evenly spaced sample) the median-of-REPS time per solve, the statistics
words, host fallbacks, and a checksum of the optima (sum of the assignments' costs in double precision: equal across exact
variants up to ties).
Usage: gpurun -- 'REART_LIB=reart_amd/csrc/libreart_hip_x.so REPS=5 python tools/replay_tail.py tools/_states/r05_tail_recipe.npz'"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reart_amd.utils import lap

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", 5))
KEEP = int(os.environ.get("KEEP", 24))          # the first KEEP dumped solves are the run's slowest, the rest the sample


def main(path):
    g = np.load(path)
    tgt_h, cols_h, prices_h = g["tgt"], g["cols"], g["prices"]
    S, B, n = g["src"].shape[:3]
    if os.environ.get("ORDER"):
        # experiment: the COLUMNS (targets) of every problem re-numbered -- ORDER=morton: along a Z-order curve (the 64 columns of a
        # wave become neighbours in space), ORDER=random: shuffled; the dumped state is re-numbered with them
        perm = np.zeros((B, n), np.int64)
        for b in range(B):
            if os.environ["ORDER"].startswith("morton"):
                q = tgt_h[b] - tgt_h[b].min(0)
                q = np.minimum((q / max(q.max(), 1e-30) * 1023).astype(np.int64), 1023)
                code = np.zeros(n, np.int64)
                for bit in range(10):
                    for ax in range(3):
                        code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
                perm[b] = np.argsort(code, kind="stable")
                if os.environ["ORDER"] == "morton2":      # two columns per thread (n = 2 x workgroup): thread t gets curve positions 2t, 2t + 1
                    perm[b] = np.concatenate([perm[b][0::2], perm[b][1::2]])
            else:
                perm[b] = np.random.default_rng(b).permutation(n)
        inv = np.zeros_like(perm)
        for b in range(B):
            inv[b, perm[b]] = np.arange(n)
        tgt_h = np.stack([tgt_h[b][perm[b]] for b in range(B)])
        prices_h = np.stack([np.stack([prices_h[s][b][perm[b]] for b in range(B)]) for s in range(S)])
        cols_h = np.stack([np.stack([np.where(cols_h[s][b] >= 0, inv[b][np.maximum(cols_h[s][b], 0)], -1) for b in range(B)]) for s in range(S)]).astype(np.int32)
    src_h = g["src"]
    if os.environ.get("ROWS"):
        # ... and the ROWS (source points) along a Z-order curve of their positions in the first dumped solve
        def morton(pts):
            q = pts - pts.min(0)
            q = np.minimum((q / max(q.max(), 1e-30) * 1023).astype(np.int64), 1023)
            code = np.zeros(len(pts), np.int64)
            for bit in range(10):
                for ax in range(3):
                    code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
            return np.argsort(code, kind="stable")
        rperm = [morton(src_h[0][b]) for b in range(B)]
        src_h = np.stack([np.stack([src_h[s][b][rperm[b]] for b in range(B)]) for s in range(S)])
        cols_h = np.stack([np.stack([cols_h[s][b][rperm[b]] for b in range(B)]) for s in range(S)])
    tgt = torch.from_numpy(tgt_h).to(dev).contiguous()
    solve = lap.InPlaceResolve(B, n, dev)
    ms, steps, cost, fbs = np.zeros((S, REPS)), np.zeros((S, B), np.int64), np.zeros(S), 0
    raws = np.zeros((S, B, 4), np.int64)
    for s in range(S):
        src = torch.from_numpy(src_h[s]).to(dev).contiguous()
        cols0, prices0 = torch.from_numpy(cols_h[s]).to(dev), torch.from_numpy(prices_h[s]).to(dev)
        for r in range(REPS + 1):               # one untimed pass first
            st = {"cols": cols0.clone(), "prices": prices0.clone()}
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fb, raw = solve(src, tgt, st, stats=True)
            b.record()
            torch.cuda.synchronize()
            if r:
                ms[s, r - 1] = a.elapsed_time(b)
            fbs += fb
        steps[s] = raw[:, 2]
        raws[s] = raw
        c = st["cols"].long()
        cost[s] = float((src.double() - torch.gather(tgt, 1, c[..., None].expand(B, n, 3)).double()).pow(2).sum(-1).sqrt().sum())
    med = np.median(ms, axis=1)
    print(f"{os.path.basename(path)}: {S} solves of {B} x {n}^2, {REPS} timed passes each; library {os.environ.get('REART_LIB', 'product')}; host fallbacks {fbs}")
    for name, sl in (("slowest of the run", slice(0, min(KEEP, S))), ("evenly spaced sample", slice(KEEP, S))):
        m = med[sl]
        if not len(m):
            continue
        print(f"  {name} ({len(m)}): sum of medians {m.sum():.2f} ms | mean {m.mean():.3f} | p50 {np.median(m):.3f} | max {m.max():.3f} | "
              f"search steps of the slowest problem, mean {steps[sl].max(1).mean():.0f} | cost checksum {cost[sl].sum():.9f}")
    print("  per solve (median ms):", " ".join(f"{v:.2f}" for v in med))
    if os.environ.get("RAW_OUT"):           # the statistics words and times per solve, for tools/replay_kernels.py
        np.savez(os.environ["RAW_OUT"] + "." + os.path.basename(path), raws=raws, med=med, reps=REPS)
    if os.environ.get("TIMES_OUT"):
        np.save(os.environ["TIMES_OUT"] + "." + os.path.basename(path) + ".npy", med)
    if os.environ.get("BUCKET_STATS"):
        # diagnostic build (make -C reart_amd/csrc phase): entries / 16 | rounds | buckets in place of the commit conflicts and the
        # reduction steps; TIMES=prefix takes the times of a product-library pass over the same dump (TIMES_OUT=prefix there)
        k, ar = steps.argmax(1), np.arange(S)
        relax = 16 * ((raws[ar, k, 1] >> 16) & 0xffff)
        arr = raws[ar, k, 3] >> 8
        rounds, buckets = arr & 0xfff, (arr >> 12) & 0xfff
        seqs = np.maximum(steps[ar, k] - rounds - buckets, 0)
        t = np.load(os.environ["TIMES"] + "." + os.path.basename(path) + ".npy") if os.environ.get("TIMES") else med
        ok = (rounds < 4000) & (buckets < 4000)
        A = np.stack([np.ones(ok.sum()), seqs[ok], rounds[ok], buckets[ok], relax[ok]], axis=1).astype(np.float64)
        c, *_ = np.linalg.lstsq(A, t[ok], rcond=None)
        print(f"  slowest problem of a solve ({ok.sum()} solves whose counters did not wrap): ms = {c[0]:.3f} + {1e3 * c[1]:.3f} us x one-column steps + "
              f"{1e3 * c[2]:.3f} us x rounds + {1e3 * c[3]:.3f} us x buckets + {1e3 * c[4]:.3f} us x entries; residual rms {(t[ok] - A @ c).std():.3f} ms; "
              f"means: steps {seqs[ok].mean():.0f} rounds {rounds[ok].mean():.0f} buckets {buckets[ok].mean():.0f} entries {relax[ok].mean():.0f}; mean ms {t[ok].mean():.3f}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
