"""Racing re-solve, measured: config-5-like problems (B = 19, n = 2048) moved a little between solves; per solve the wall
time of the plain re-solve and of the raced one (from the SAME start state), and which racer won.
    python tools/lap_race_exp.py [racers ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from reart_amd.utils import lap  # noqa: E402

dev = torch.device("cuda:0")
B, n = 19, 2048
rng = np.random.default_rng(5)
a = rng.uniform(-0.3, 0.3, (B, n, 3)).astype(np.float32)
b = (a[:, rng.permutation(n)] + rng.normal(0, 0.004, (B, n, 3))).astype(np.float32)
tb = torch.from_numpy(b).to(dev)
variants = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 12]
state = {}
lap.linear_sum_assignment_points(torch.from_numpy(a).to(dev), tb, state)
wins = {r: np.zeros(16, np.int64) for r in variants}
tot = {r: [] for r in variants}
for step in range(12):
    a = (a + rng.normal(0, 0.0015, a.shape)).astype(np.float32)
    ta = torch.from_numpy(a).to(dev)
    nxt = None
    for r in variants:
        st = {k: v.clone() for k, v in state.items()}
        lap.RESOLVE_RACERS = r
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        _, fb, stats = lap.linear_sum_assignment_points(ta, tb, st, return_stats="full", race=r > 1)
        e1.record()
        torch.cuda.synchronize()
        tot[r].append(e0.elapsed_time(e1))
        for w in stats[:, 0] >> 16:
            wins[r][w] += 1
        assert fb == 0
        if nxt is None:
            nxt = st
    state = nxt
for r in variants:
    print(f"racers {r:2d}: {np.mean(tot[r]):8.3f} ms per re-solve (incl. host copy-back)  winners {wins[r][:r].tolist()}")
