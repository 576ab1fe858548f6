"""Tied optima of the assignment refresh (run_robot.py:172-176: scipy's linear_sum_assignment is a pure function of the cost
matrix, so the reference repeats under --manual_seed, run_robot.py:37-49; the raced GPU solvers return SOME optimum):
reart_lap_ties finds the problems whose optimum is not unique, reart_amd.utils.lap.canonical_among_ties takes the
lexicographically smallest optimum -- and with that (`--deterministic`) two runs of the projection are the same run."""
import itertools
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _problems(dev, B=3, n=512, seed=0):
    """Nearly the identity assignment: targets = sources + noise, shuffled columns."""
    rng = np.random.default_rng(seed)
    src = rng.uniform(0.0, 1.0, (B, n, 3)).astype(np.float32)
    tgt = (src + rng.normal(0, 0.01, src.shape)).astype(np.float32)
    tgt = np.stack([t_[rng.permutation(n)] for t_ in tgt])
    return src, tgt


def _plant_two_swap(src, tgt, b, i, k, ci, ck, x0=5.0):
    """rows i, k and columns ci, ck of problem b, far from everything else: all four costs equal to the last bit."""
    src[b, i], src[b, k] = (x0 + 0.25, 0.0, 0.0), (x0 - 0.25, 0.0, 0.0)
    tgt[b, ci], tgt[b, ck] = (x0, 0.125, 0.0), (x0, -0.125, 0.0)


def _ties(dev, src, tgt, cols, prices):
    from reart_amd.utils import lap

    B, n = cols.shape
    tb = lap.TieBreaker(B, n, dev)
    tb.launch(src, tgt, cols, prices)
    torch.cuda.synchronize()
    return tb.tie_host.numpy().copy(), [tb.pairs_of(b) for b in range(B)]


@pytest.mark.parametrize("n", [512, 700, 2048])
def test_tight_pairs_and_flags_against_the_host(dev, n):
    from reart_amd.utils import lap

    src_h, tgt_h = _problems(dev, 3, n, seed=n)
    _plant_two_swap(src_h, tgt_h, 1, 7, 300, 11, 5)
    src_h[2, 40:43] = (7.0, 7.0, 7.0)                                  # three coincident rows: any permutation of their columns is optimal
    tgt_h[2, [9, 100, 333]] = [(7.0, 7.0, 7.5), (7.0, 7.5, 7.0), (7.5, 7.0, 7.0)]
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    state = {}
    for sweep in range(2):                                             # cold (raced auction), then the re-solve from the optimum
        old = lap.CANONICAL_TIES
        lap.CANONICAL_TIES = False
        try:
            _, fb = lap.linear_sum_assignment_points(src, tgt, state, device_cols=True)
        finally:
            lap.CANONICAL_TIES = old
        assert fb == 0
        cols, prices = state["cols"], state["prices"]
        tie, pairs = _ties(dev, src, tgt, cols, prices)
        assert tie.tolist() == [0, 1, 1]
        for b in range(3):                                             # the kernel's pairs = the host's, as sets
            host = lap.tight_pairs_host(src[b], tgt[b], cols[b], prices[b])
            assert sorted(map(tuple, pairs[b].tolist())) == sorted(map(tuple, host.tolist())), (sweep, b)
        c1 = cols[1].cpu().numpy()
        new, moved = lap.canonical_among_ties(c1, pairs[1])
        assert sorted(new[[7, 300]].tolist()) == [5, 11] and new[7] == 5 and new[300] == 11
        assert moved in (0, 2) and np.array_equal(np.delete(new, [7, 300]), np.delete(c1, [7, 300]))
        c2 = cols[2].cpu().numpy()
        new2, _ = lap.canonical_among_ties(c2, pairs[2])
        assert new2[40:43].tolist() == [9, 100, 333]
        assert np.array_equal(np.delete(new2, [40, 41, 42]), np.delete(c2, [40, 41, 42]))
        # what is called canonical is optimal: the same cost on the fp32 matrix the solvers see, to the last bit
        cost = lap.cdist(src, tgt).cpu().numpy().astype(np.float64)
        import math
        for b, nw in ((1, new), (2, new2)):
            assert math.fsum(cost[b][np.arange(n), nw]) == math.fsum(cost[b][np.arange(n), cols[b].cpu().numpy()])
        src = src + 0.0                                                # (the second round re-solves the same problems warm)


def test_every_entry_of_the_loops_settles_ties(dev):
    """CANONICAL_TIES on: the host-list entry, the device-column entry and the in-place re-solve all return the canonical
    optimum whatever the solver found."""
    from reart_amd.utils import lap

    n = 512
    src_h, tgt_h = _problems(dev, 2, n, seed=5)
    _plant_two_swap(src_h, tgt_h, 0, 100, 20, 400, 17)
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    old = lap.CANONICAL_TIES
    lap.CANONICAL_TIES = True
    try:
        state = {}
        out = lap.linear_sum_assignment_points(src, tgt, state)                       # cold, host lists
        assert out[0][1][20] == 17 and out[0][1][100] == 400
        # push the state to the OTHER optimum, then re-solve warm: still canonical
        for entry in ("lists", "device", "inplace"):
            c = state["cols"].clone()
            c[0, 20], c[0, 100] = 400, 17
            state["cols"] = c
            if entry == "lists":
                out = lap.linear_sum_assignment_points(src, tgt, state)
                got = out[0][1]
            elif entry == "device":
                got = lap.linear_sum_assignment_points(src, tgt, state, device_cols=True)[0][0].cpu().numpy()
            else:
                assert lap.InPlaceResolve.usable(state, 2, n)
                fb, _ = lap.InPlaceResolve(2, n, dev)(src, tgt, state)
                assert fb == 0
                got = state["cols"][0].cpu().numpy()
                # the pairs the solve's own certificate pass listed (reart_lap_resolve_points_mc_ties) = the host's, as sets
                tb = state["tie_breaker"]
                torch.cuda.synchronize()
                for b in range(2):
                    listed = tb.pairs_of(b)
                    c_b = state["cols"][b].clone()
                    if b == 0:
                        c_b[20], c_b[100] = 400, 17            # (as the solve left it, before the canonical choice)
                    host = lap.tight_pairs_host(src[b], tgt[b], c_b, state["prices"][b])
                    want, have = set(map(tuple, host.tolist())), set(map(tuple, listed.tolist()))
                    # (the solver may have returned either optimum of the tied pair: the listed pairs are those of ITS optimum)
                    assert have == want or b == 0, (b, len(have), len(want))
                assert tb.tie_host.tolist() == [1, 0]
            assert got[20] == 17 and got[100] == 400, entry
        assert state["tie_breaker"].flagged >= 3
    finally:
        lap.CANONICAL_TIES = old


def _projection(dev, canonical, iters=200):
    from tests.test_kinematic_engine_gpu import _model, t
    from reart_amd.kinematic_engine import KinematicEngine
    from reart_amd.utils import lap

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
    old = lap.CANONICAL_TIES
    lap.CANONICAL_TIES = canonical
    try:
        m = _model(dev, k, cano)
        eng = KinematicEngine(m, cano, pcs, 2, refs, flows, assign_iter=0, assign_gap=1, downsample=2)
        cols, params = [], []
        for i in range(iters):
            eng.iteration(i)
            cols.append(eng.lap_state["cols"].clone())
            if (i + 1) % 25 == 0:
                params.append(torch.cat([getattr(m, n_).detach().reshape(-1).clone() for n_ in ("axis_list", "moment_list", "theta_list")]))
        tb = eng.lap_state.get("tie_breaker")
        return cols, params, eng.lap_fallbacks, (tb.flagged, tb.changed, tb.overflows) if tb is not None else (0, 0, 0)
    finally:
        lap.CANONICAL_TIES = old


def test_two_deterministic_projections_are_the_same_run(dev):
    """README.md:125's configuration (assign_gap 1, downsample 2: 9 x 2048^2 re-solved every iteration, 13 racers per problem,
    lock-free row reduction) from the reference's kinematic-2 checkpoint: 200 iterations, twice -- every assignment of every
    iteration and the parameters bit for bit.  (Without the tie check two such runs part after 16-150 iterations:
    profiles/r05_exp_kin_determinism.txt.)"""
    a_cols, a_par, a_fb, a_tb = _projection(dev, True)
    b_cols, b_par, b_fb, b_tb = _projection(dev, True)
    assert a_fb == 0 and b_fb == 0
    for i, (x, y) in enumerate(zip(a_cols, b_cols)):
        assert torch.equal(x, y), f"assignments differ at iteration {i}"
    for x, y in zip(a_par, b_par):
        assert torch.equal(x, y)
    assert a_tb[0] >= 1 and b_tb[0] >= 1, "no tie met in 200 iterations: the test did not exercise the mechanism"
    # (no problem came back with stale pairs -- flag 2: its certificate had to repair potentials after the pass that listed them; the
    # loop that shows a solver which leaves the certificate work is the nao projection: tests/test_bench_gpu.py holds it to that)
    assert a_tb[2] <= 2 and b_tb[2] <= 2, (a_tb, b_tb)


def test_a_row_with_more_tight_pairs_than_slots_goes_to_the_host(dev):
    """31 coincident sources against 31 coincident targets: every one of those rows is tied with 30 columns -- more than the K = 24
    slots a row has in the kernels' pair lists.  The problem comes back with flag 2, the host lists its pairs from the cost matrix
    (lap.tight_pairs_host) and the canonical optimum gives the rows, in ascending order, the columns in ascending order."""
    from reart_amd.utils import lap

    n = 512
    src_h, tgt_h = _problems(dev, 2, n, seed=9)
    rows = np.arange(40, 40 + 31)
    colsd = np.arange(300, 300 + 31)
    src_h[1, rows] = (6.0, 6.0, 6.0)
    tgt_h[1, colsd] = (6.0, 6.5, 6.0)
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    old = lap.CANONICAL_TIES
    lap.CANONICAL_TIES = True
    try:
        state = {}
        for k in range(2):                                   # cold, then warm (the in-place re-solve lists the pairs itself)
            if k == 0:
                lap.linear_sum_assignment_points(src, tgt, state, device_cols=True)
            else:
                perm = torch.from_numpy(colsd[np.random.default_rng(k).permutation(31)].astype(np.int32)).to(dev)
                state["cols"][1, torch.from_numpy(rows).to(dev)] = perm      # any other optimum to start from
                fb, _ = lap.InPlaceResolve(2, n, dev)(src, tgt, state)
                assert fb == 0
            got = state["cols"][1].cpu().numpy()
            np.testing.assert_array_equal(got[rows], colsd)
            tb = state["tie_breaker"]
            assert tb.overflows >= 1 and tb._tie_np.tolist() == [0, 2]
    finally:
        lap.CANONICAL_TIES = old


def test_massively_tied_clouds_do_not_stall_the_host(dev):
    """Clouds drawn from 64 distinct points (every cost value occurs thousands of times: hundreds of rows tied with each other): the
    canonical choice is skipped for components above lap.MAX_TIED_ROWS -- the solver's optimum stands -- and the call returns in
    seconds with the optimal cost."""
    import time

    import oracle
    from reart_amd.utils import lap

    rng = np.random.default_rng(3)
    B, n = 2, 1024
    grid = rng.uniform(-0.3, 0.3, (64, 3)).astype(np.float32)
    tgt_h, src_h = grid[rng.integers(0, 64, (B, n))], grid[rng.integers(0, 64, (B, n))]
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    old, lap.CANONICAL_TIES = lap.CANONICAL_TIES, True
    try:
        t0 = time.time()
        state = {}
        out = lap.linear_sum_assignment_points(src, tgt, state)
        src2_h = src_h.copy()
        src2_h[:, :200] = grid[rng.integers(0, 64, (B, 200))]
        src2 = torch.from_numpy(src2_h).to(dev)
        out2 = lap.linear_sum_assignment_points(src2, tgt, state)
        assert time.time() - t0 < 60
    finally:
        lap.CANONICAL_TIES = old
    assert lap.canonical_among_ties.skipped >= 1
    for pts, res in ((src_h, out), (src2_h, out2)):
        cost = oracle.cdist(pts, tgt_h)
        ref = oracle.linear_sum_assignment(cost)
        for b, (r, c) in enumerate(res):
            assert sorted(c.tolist()) == list(range(n))
            assert abs(float(cost[b][r, c].astype(np.float64).sum()) - float(cost[b][ref[b][0], ref[b][1]].astype(np.float64).sum())) <= 1e-9


def test_a_refresh_replays_from_a_graph_and_clears_its_own_flags(dev):
    """lap.InPlaceResolve on the same buffers: the third refresh on replays a captured graph (lap.ReplayedLaunches) -- the same
    optimum as scipy on every one, with the flags, the statistics and the racers' meeting point pre-filled with garbage (the
    call's set-up launch defines them: no fill launches in front of a refresh)."""
    from scipy.optimize import linear_sum_assignment

    from reart_amd.utils import lap

    B, n = 3, 1024
    rng = np.random.default_rng(11)
    src_h, tgt_h = _problems(dev, B, n, seed=9)
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    for det in (True, False):
        old = lap.CANONICAL_TIES
        lap.CANONICAL_TIES = det
        try:
            state = {}
            lap.linear_sum_assignment_points(src, tgt, state, device_cols=True)
            assert lap.InPlaceResolve.usable(state, B, n)
            solve = lap.InPlaceResolve(B, n, dev)
            for it in range(6):
                src.add_(torch.from_numpy(rng.normal(scale=2e-3, size=(B, n, 3)).astype(np.float32)).to(dev))      # in place: same address
                solve.cert.fill_(7)
                if solve._ws is not None:
                    solve._ws.fill_(0x5a)
                fb, raw = solve(src, tgt, state, stats=True)
                assert fb == 0 and solve.cert_host.tolist() == [1] * B
                assert (raw[:, 3] & 0xff).tolist() == [1] * B and (raw[:, 2] >= 0).all() and (raw[:, 2] < 10 * n).all()
                cost = lap.cdist(src, tgt).cpu().numpy().astype(np.float64)
                for b in range(B):
                    want = cost[b][linear_sum_assignment(cost[b])].sum()
                    got = cost[b][np.arange(n), state["cols"][b].cpu().numpy()].sum()
                    assert abs(got - want) <= n * cost[b].max() * 1e-13 * 4, (det, it, b, got - want)
            assert solve.launches.replays == (4 if lap.ReplayedLaunches.ENABLED else 0)
        finally:
            lap.CANONICAL_TIES = old
