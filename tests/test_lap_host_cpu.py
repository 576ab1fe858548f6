"""Host logic of reart_amd.utils.lap around the GPU solver (no GPU, no compute): what happens when the kernels do NOT
certify a matrix.  The library is replaced by a fake whose solver entry points return success and write nothing -- exactly
what a racing launch leaves behind for a matrix on which every racer missed the certificate (lap.hip: nobody publishes)."""
import numpy as np
import pytest
import torch


class _FakeLib:
    """reart_lap_* entry points that succeed without touching their outputs; optionally certifies some matrices with a
    given assignment (written through the raw pointers like the kernel would)."""

    def __init__(self, B, n, certify=None):
        self.B, self.n, self.certify, self.calls = B, n, certify or {}, []
        # like a ctypes library, attribute access returns the SAME function object every time (lap.py compares them by identity)
        for name in [k for k in dir(type(self)) if k.startswith("reart_")]:
            setattr(self, name, getattr(self, name))

    def reart_lap_workspace_bytes(self, B, n):
        return 8 * B * n + 4096 + 16 * B + 8 * B * n

    def reart_lap_race_workspace_bytes(self, B, n, racers):
        return self.reart_lap_workspace_bytes(B, n) * racers

    def _solve(self, name, col_ptr, cert_ptr):
        import ctypes

        self.calls.append(name)
        for b, cols in self.certify.items():
            ctypes.memmove(col_ptr.value + 4 * b * self.n, np.asarray(cols, np.int32).ctypes.data, 4 * self.n)
            ctypes.memmove(cert_ptr.value + 4 * b, np.asarray([1], np.int32).ctypes.data, 4)
        return 0

    def reart_lap_auction(self, cost, B, n, col, cert, pin, pout, ws, nws, st):
        return self._solve("auction", col, cert)

    def reart_lap_auction_race(self, cost, src, tgt, B, n, racers, col, cert, pout, ws, nws, st):
        return self._solve("race", col, cert)

    def reart_lap_auction_race_warm(self, cost, src, tgt, B, n, racers, cin, pin, col, cert, pout, ws, nws, st):
        return self._solve("race_warm", col, cert)

    def reart_lap_auction_warm(self, *a):
        return self._solve("warm", a[3], a[4])

    def reart_lap_resolve(self, *a):
        return self._solve("resolve", a[3], a[4])

    def reart_lap_auction_points(self, cost, src, tgt, B, n, col, cert, pin, pout, ws, nws, st):
        return self._solve("points", col, cert)

    def reart_status_string(self, rc):
        return b"ok"


@pytest.fixture
def fake(monkeypatch):
    from reart_amd import _lib

    def install(B, n, certify=None):
        f = _FakeLib(B, n, certify)
        monkeypatch.setattr(_lib, "lib", lambda: f)
        monkeypatch.setattr(_lib, "require_gpu", lambda *t: None)
        monkeypatch.setattr(_lib, "stream", lambda: None)
        monkeypatch.setattr(_lib, "workspace", lambda nbytes, device: torch.full((max(int(nbytes), 1),), 0xAB, dtype=torch.uint8))
        return f

    return install


def _optimal(cost):
    from scipy.optimize import linear_sum_assignment

    return [linear_sum_assignment(c)[1] for c in cost.numpy()]


@pytest.mark.parametrize("race", [True, "warm", False])
def test_uncertified_matrices_are_solved_on_the_host_and_leave_no_garbage(fake, race):
    from reart_amd.utils import lap

    B, n = 3, 12
    rng = np.random.default_rng(0)
    cost = torch.from_numpy(rng.uniform(0, 1, (B, n, n)).astype(np.float32))
    want = _optimal(cost)
    f = fake(B, n, certify={1: want[1]})                   # the kernels certify matrix 1 only
    state = {}
    out, fallbacks, stats = lap.linear_sum_assignment_batch(cost, return_stats="full", state=state, race=race)
    assert fallbacks == 2 and f.calls[0] in ("race", "auction")
    for b in range(B):                                     # always the optimum scipy returns
        np.testing.assert_array_equal(out[b][0], np.arange(n))
        np.testing.assert_array_equal(out[b][1], want[b])
    assert (stats == 0).all()                              # no statistics were written: zeros, not the buffer's old bytes
    # what is kept for the next call: finite, defined -- zero potentials for the matrices the host solved
    assert torch.isfinite(state["prices"]).all()
    assert (state["prices"][0] == 0).all() and (state["prices"][2] == 0).all()
    if race == "warm":
        np.testing.assert_array_equal(state["cols"].numpy(), np.stack(want))      # the host's optimum, not uninitialised memory
        # second call: the warm racers start from that state
        out2 = lap.linear_sum_assignment_batch(cost, state=state, race="warm")
        assert f.calls[-1] == "race_warm"
        for b in range(B):
            np.testing.assert_array_equal(out2[b][1], want[b])
        assert torch.isfinite(state["prices"]).all()
        np.testing.assert_array_equal(state["cols"].numpy(), np.stack(want))


def test_points_are_validated_in_every_branch(fake):
    from reart_amd.utils import lap

    B, n = 2, 8
    cost = torch.rand(B, n, n)
    fake(B, n)
    with pytest.raises(ValueError):
        lap.linear_sum_assignment_batch(cost, points=(torch.rand(B, n - 1, 3), torch.rand(B, n, 3)), race=True)
    with pytest.raises(ValueError):
        lap.linear_sum_assignment_batch(cost, points=(torch.rand(B, n, 3), torch.rand(B, n, 2)), race=False)


def test_racing_branch_requires_gpu_points():
    """The real require_gpu: host tensors never reach a kernel as pointers."""
    from reart_amd.utils import lap

    with pytest.raises(RuntimeError):
        lap.linear_sum_assignment_batch(torch.rand(1, 4, 4), race=True)


def test_resolve_racer_count():
    """Host policy of the raced re-solve: a racer is one workgroup holding a compute unit's LDS, the chip has 256; short
    problems (one launch) and batches that leave no idle unit are not raced."""
    from reart_amd.utils import lap

    assert lap._resolve_racers(19, 2048) == min(lap.RESOLVE_RACERS, 13)
    assert lap._resolve_racers(19, 2048, race=False) == 1
    assert lap._resolve_racers(19, 511) == 1
    assert lap._resolve_racers(100, 1024) == 2
    assert lap._resolve_racers(129, 1024) == 1
    assert lap._resolve_racers(1, 2048) == lap.RESOLVE_RACERS
    assert 2 <= lap.RESOLVE_RACERS <= 13 or lap.RESOLVE_RACERS == 1


def test_spatial_order_is_a_permutation_that_keeps_neighbours_together():
    """lap.spatial_order (the loops number their target columns with it): per problem a permutation, the same on every call, and
    consecutive points along it are neighbours in space (a Z-order curve) -- which is all the solver's speed needs; the optimum
    does not depend on the numbering."""
    import numpy as np
    import torch
    from reart_amd.utils.lap import spatial_order

    rng = np.random.default_rng(0)
    pts = torch.from_numpy(rng.normal(size=(3, 2048, 3)).astype(np.float32))
    pts[2] *= 1e-3                                       # a tiny cloud is scaled to the same grid
    order = spatial_order(pts)
    assert order.shape == (3, 2048) and order.dtype == torch.long
    assert torch.equal(order, spatial_order(pts))
    for b in range(3):
        assert torch.equal(torch.sort(order[b]).values, torch.arange(2048))
        along = pts[b][order[b]]
        hop = (along[1:] - along[:-1]).norm(dim=-1).mean()
        rand = (pts[b][1:] - pts[b][:-1]).norm(dim=-1).mean()
        assert hop < 0.25 * rand, (b, float(hop), float(rand))
    # degenerate input: all points equal -> the identity (stable sort of equal codes)
    same = torch.zeros((1, 16, 3))
    assert torch.equal(spatial_order(same)[0], torch.arange(16))


def test_canonical_among_ties_is_the_lexicographic_minimum_of_all_optima():
    """reart_amd.utils.lap.canonical_among_ties (the host half of --deterministic; reference: scipy's refresh is a function of
    the cost matrix, run_robot.py:172-176): small integer matrices with many tied optima; starting from a RANDOM optimal
    assignment and optimal potentials built for it, the result is the lexicographically smallest of all optimal permutations
    (brute force) -- whatever the start."""
    import itertools

    from scipy.optimize import linear_sum_assignment

    from reart_amd.utils.lap import canonical_among_ties

    rng = np.random.default_rng(0)
    tied = 0
    for trial in range(1500):
        n = int(rng.integers(2, 7))
        C = rng.integers(0, 4, (n, n)).astype(np.float64)
        r, c = linear_sum_assignment(C)
        opt = C[r, c].sum()
        opts = [p for p in itertools.permutations(range(n)) if C[np.arange(n), list(p)].sum() == opt]
        tied += len(opts) > 1
        start = np.array(opts[int(rng.integers(len(opts)))])
        p = np.zeros(n)                                     # prices with c_ij + p_j >= c_i,s(i) + p_s(i): longest paths
        for _ in range(n + 2):
            for i in range(n):
                for j in range(n):
                    p[j] = max(p[j], p[start[i]] + C[i, start[i]] - C[i, j])
        red = C + p[None, :]
        cur = red[np.arange(n), start]
        assert (red - cur[:, None] >= 0).all()
        hit = (red - cur[:, None]) <= 0
        hit[np.arange(n), start] = False
        new, moved = canonical_among_ties(start, np.argwhere(hit))
        assert tuple(new) == opts[0], (C, start, new, opts[0])
        assert moved == int((new != start).sum())
    assert tied > 300


def test_replayed_launches_state_machine(monkeypatch):
    """lap.ReplayedLaunches (the refresh's launches replayed from a captured graph): eager until the same key has come twice,
    captured once by settle() -- not while a host fallback touched the state, not when switched off --, replayed from then on,
    dropped when the key changes; the guard is entered around the capture only.  The graph objects are stand-ins: the logic is
    the host's."""
    from reart_amd.utils import lap

    events = []

    class FakeGraph:
        def replay(self):
            events.append("replay")

    class FakeCapture:
        def __init__(self, g, **kw):
            pass

        def __enter__(self):
            events.append("capture-begin")

        def __exit__(self, *exc):
            events.append("capture-end")

    monkeypatch.setattr(torch.cuda, "CUDAGraph", FakeGraph)
    monkeypatch.setattr(torch.cuda, "graph", FakeCapture)
    monkeypatch.setattr(lap.ReplayedLaunches, "ENABLED", True)
    fn = lambda: events.append("queue")

    class Guard:
        def __enter__(self):
            events.append("guard-in")

        def __exit__(self, *exc):
            events.append("guard-out")

    r = lap.ReplayedLaunches()
    r.guard = Guard
    r.run(("a", 1), fn); r.settle(fn)                       # first time: eager, too early to capture
    assert events == ["queue"] and r.graph is None
    r.run(("a", 1), fn); r.settle(fn, ok=False)             # second time, but the host touched the state: put off
    assert events == ["queue", "queue"] and r.graph is None
    r.run(("a", 1), fn); r.settle(fn)                       # captured AFTER the refresh it would have served (the capture queues nothing real)
    assert events[2:] == ["queue", "guard-in", "capture-begin", "queue", "capture-end", "guard-out"] and r.graph is not None
    del events[:]
    r.run(("a", 1), fn); r.settle(fn)
    r.run(("a", 1), fn); r.settle(fn)
    assert events == ["replay", "replay"] and r.replays == 2
    r.run(("b", 1), fn); r.settle(fn)                       # another buffer / launch parameter: eager again, the old graph is gone
    assert events[2:] == ["queue"] and r.graph is None and r.seen == 1
    monkeypatch.setattr(lap.ReplayedLaunches, "ENABLED", False)
    del events[:]
    for _ in range(4):
        r.run(("b", 1), fn); r.settle(fn)
    assert events == ["queue"] * 4 and r.graph is None
