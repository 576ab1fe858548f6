"""Batched linear assignment for the assignment loss (reference run_robot.py:164-187,
utils/model_utils.py:85-103): ``[linear_sum_assignment(c) for c in cost]`` / ``parallel_lap(cost, nproc)``
on the GPU (``reart_lap_auction``: epsilon-scaling auction + exact dual certificate).  A matrix whose
certificate does not close is solved with scipy on the host, so the result is always an optimal assignment."""
import contextlib
import os

import numpy as np
import torch

from .. import _lib


def cdist(a, b):
    """``torch.cdist(a, b)`` for point batches a [B,n,3], b [B,m,3] -> [B,n,m] Euclidean distances by direct
    differences (run_robot.py:171, utils/model_utils.py:93): one pass over the output."""
    _lib.require_gpu(a, b)
    a, b = a.detach().contiguous().float(), b.detach().contiguous().float()
    if a.dim() != 3 or b.dim() != 3 or a.shape[0] != b.shape[0] or a.shape[2] != 3 or b.shape[2] != 3:
        raise ValueError("cdist expects [B,n,3] and [B,m,3]")
    out = torch.empty((a.shape[0], a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device)
    rc = _lib.lib().reart_cdist(_lib.ptr(a), _lib.ptr(b), a.shape[0], a.shape[1], b.shape[1], _lib.ptr(out), _lib.stream())
    _lib.check(rc, "reart_cdist")
    return out


RACERS = 5   # epsilon schedules raced per matrix by ``race=True`` (reart_lap_auction_race) at n = 4096


def _racers(B, n, warm):
    """(cold racers, all racers) per matrix: the chip has 256 compute units and a racer is one workgroup.  Up to 2048 columns
    more schedules keep paying (12: -6 % at 1024^2, -10 % at 2048^2 against 5); at 4096^2 the racers' matrix reads start to
    contend (+3 %).  ``warm``: three of the racers start from the previous potentials / assignment."""
    if n > 2048:
        return RACERS, RACERS + (3 if warm else 0)
    total = max(RACERS, min(15 if warm else 12, 256 // max(B, 1)))
    return (max(total - 3, 2), total) if warm else (total, total)


def linear_sum_assignment_batch(cost, return_stats=False, state=None, warm_assignment=False, method="paths", points=None,
                                race=False):
    """cost [B,n,n] float32 CUDA tensor (square) -> list of (row_ind, col_ind) int64 numpy arrays, like
    ``[scipy.optimize.linear_sum_assignment(c) for c in cost]`` (rows in ascending order).
    ``state``: a dict kept by the caller between calls on slowly changing matrices (the loop re-solves every
    ``assign_gap`` iterations); it carries the column potentials of the previous solve as a warm start.
    ``warm_assignment=True`` (with ``state``) also carries the previous assignment and keeps the pairs that are still
    tight (``reart_lap_auction_warm``): faster when the matrices move smoothly (KinematicModel), slower when they
    jump (BaseModel's resampled labels) -- there, and by default, solve cold (``state=None``).
    ``method`` of the warm re-solve: "paths" = shortest augmenting paths from the previous assignment and potentials
    (``reart_lap_resolve``), "auction" = the warm-started auction (``reart_lap_auction_warm``).
    ``points=(src, tgt)`` ([B,n,3] each) when ``cost`` is ``cdist(src, tgt)``: a cold solve then recomputes the rows of its
    long single-bidder chains from the points instead of reading them (``reart_lap_auction_points``; same result).
    ``race=True`` (cold solves): five workgroups per matrix race with different epsilon schedules on otherwise idle compute
    units and the first certified one publishes (``reart_lap_auction_race``): the same optimal assignment 20 % sooner; the
    potentials kept in ``state`` are the winner's (valid, but not reproducible run to run; when a matrix has several optimal
    assignments of exactly the same cost, which of them is returned may also differ from run to run -- the winner's).
    ``race="warm"`` with a ``state``
    kept between calls: from the second call on three more racers start from the previous potentials and assignment
    (``reart_lap_auction_race_warm``) -- a loop need not know whether its matrices moved little or jumped."""
    _lib.require_gpu(cost)
    if cost.dim() != 3 or cost.shape[1] != cost.shape[2]:
        raise ValueError("linear_sum_assignment_batch expects square matrices [B,n,n]")
    cost = cost.detach().float().contiguous()
    B, n, _ = cost.shape
    L = _lib.lib()
    src = tgt = None
    if points is not None:      # validated for every branch that may hand the pointers to a kernel
        src, tgt = (p.detach().float().contiguous() for p in points)
        _lib.require_gpu(src, tgt)
        if tuple(src.shape) != (B, n, 3) or tuple(tgt.shape) != (B, n, 3):
            raise ValueError("points = (src, tgt), both [B,n,3], with cost = cdist(src, tgt)")
        if src.device != cost.device or tgt.device != cost.device:
            raise ValueError("points and cost live on different devices")
    # defined outputs whatever the kernels write: a racing launch whose racers all miss the certificate writes nothing for
    # that matrix (it is solved on the host below)
    col = torch.full((B, n), -1, dtype=torch.int32, device=cost.device)
    cert = torch.zeros((B,), dtype=torch.int32, device=cost.device)
    nbytes = L.reart_lap_workspace_bytes(B, n)
    if nbytes == 0:   # n > 4096: beyond the kernel's LDS state -- the reference's host solver
        from scipy.optimize import linear_sum_assignment

        out = [linear_sum_assignment(c) for c in cost.cpu().numpy()]
        if return_stats == "full":
            return out, B, np.zeros((B, 4), np.int32)
        return (out, B) if return_stats else out
    prices, warm, keep = None, False, False
    if state is not None:
        prices = state.get("prices")
        warm = prices is not None and tuple(prices.shape) == (B, n) and prices.device == cost.device
        if not warm:
            prices = torch.zeros((B, n), dtype=torch.float64, device=cost.device)
        state["prices"] = prices
        keep = bool(warm_assignment and warm and state.get("cols") is not None and tuple(state["cols"].shape) == (B, n))
        if keep:
            col.copy_(state["cols"])
    solve = L.reart_lap_auction
    if state is not None and keep:
        solve = L.reart_lap_resolve if method == "paths" else L.reart_lap_auction_warm
    racing = bool(race) and solve is L.reart_lap_auction
    # race="warm": with the potentials and assignment of the previous call in ``state`` three more racers start from them
    race_warm = racing and race == "warm" and warm and state.get("cols") is not None and tuple(state["cols"].shape) == (B, n)
    if racing and warm and not race_warm:
        racing = False                                                         # warm potentials only: the plain warm auction
    n_cold, n_racers = _racers(B, n, race_warm)
    ws = _lib.workspace(L.reart_lap_race_workspace_bytes(B, n, n_racers) if racing else nbytes, cost.device)
    if return_stats == "full":      # the statistics of a matrix nobody certified in a race are never written: report zeros
        off = ((8 * B * n + 255) // 256) * 256
        ws[off:off + 16 * B].zero_()
    tail_args = (B, n, _lib.ptr(col), _lib.ptr(cert), _lib.ptr(prices) if (state is not None and warm) else None,
                 _lib.ptr(prices) if state is not None else None, _lib.ptr(ws), ws.numel(), _lib.stream())
    if racing:
        if race_warm:
            new_prices = torch.zeros_like(prices)
            rc = L.reart_lap_auction_race_warm(_lib.ptr(cost), _lib.ptr(src), _lib.ptr(tgt), B, n, n_racers, _lib.ptr(state["cols"]),
                                               _lib.ptr(prices), _lib.ptr(col), _lib.ptr(cert), _lib.ptr(new_prices), _lib.ptr(ws),
                                               ws.numel(), _lib.stream())
            state["prices"] = new_prices
        else:
            rc = L.reart_lap_auction_race(_lib.ptr(cost), _lib.ptr(src), _lib.ptr(tgt), B, n, n_cold, _lib.ptr(col), _lib.ptr(cert),
                                          _lib.ptr(prices) if state is not None else None, _lib.ptr(ws), ws.numel(), _lib.stream())
    elif points is not None and solve is L.reart_lap_auction:
        rc = L.reart_lap_auction_points(_lib.ptr(cost), _lib.ptr(src), _lib.ptr(tgt), *tail_args)
    else:
        rc = solve(_lib.ptr(cost), *tail_args)
    _lib.check(rc, "reart_lap_auction")
    col_h, cert_h = col.cpu().numpy().astype(np.int64), cert.cpu().numpy()
    rows = np.arange(n, dtype=np.int64)
    out, fallbacks = [], 0
    for b in range(B):
        if cert_h[b]:
            out.append((rows, col_h[b]))
        else:  # certificate did not close: exact host solve for this matrix
            from scipy.optimize import linear_sum_assignment

            fallbacks += 1
            out.append(linear_sum_assignment(cost[b].cpu().numpy()))
            _forget_uncertified(state, col, b, out[-1][1])
    if state is not None and (warm_assignment or race == "warm"):
        state["cols"] = col.clone()
    if return_stats == "full":   # per matrix: phases, auction rounds, bids, certificate rounds
        off = ((8 * B * n + 255) // 256) * 256
        st = ws[off:off + 16 * B].view(torch.int32).reshape(B, 4).cpu().numpy()
        return out, fallbacks, st
    return (out, fallbacks) if return_stats else out


def _forget_uncertified(state, col, b, host_cols):
    """Matrix b was solved on the host: what the kernel left for it (an uncertified assignment and potentials, or nothing
    at all after a lost race) must not start the next solve.  The kept assignment becomes the host's optimum and the
    potentials of that matrix are zeroed -- the next call starts cold for it (useless potentials are legal input)."""
    col[b] = torch.from_numpy(np.asarray(host_cols)).to(device=col.device, dtype=col.dtype)
    if state is not None and state.get("prices") is not None:
        state["prices"][b].zero_()


# --deterministic (run_robot.py): when several assignments are optimal -- exact ties of the fp32 costs, about one re-solve in a
# few hundred of the kinematic projection -- the raced solvers return whichever the winner found; with this switch every solve
# of the loops is followed by reart_lap_ties and a tied problem takes the lexicographically smallest optimum
# (canonical_among_ties): the assignment becomes a function of the cost matrix alone, like the reference's scipy call
# (run_robot.py:172-176), and a run repeats under --manual_seed (run_robot.py:37-49)
CANONICAL_TIES = os.environ.get("REART_CANONICAL_TIES", "0") == "1"


def _lex_min_matching(rows, allowed, cols):
    """Lexicographically smallest perfect matching (rows ascending, each the smallest column it can get) of the bipartite
    graph ``allowed[row] = set of columns``, starting from the perfect matching ``cols[row]``.  Rows are fixed in ascending
    order: row r takes candidate c iff the row that holds c can then be re-matched along an alternating path through rows
    and columns not fixed yet (Kuhn's augmenting step; the matching stays perfect throughout)."""
    match = {int(r): int(cols[r]) for r in rows}
    holder = {c: r for r, c in match.items()}
    fixed_rows, fixed_cols = set(), set()
    for r in sorted(match):
        for c in sorted(allowed[r]):
            if c in fixed_cols:
                continue
            if match[r] == c:
                break
            saved = (dict(match), dict(holder))
            r2, old = holder[c], match[r]                 # r takes c: r2 loses it, r's old column becomes free
            match[r], holder[c] = c, r
            del holder[old], match[r2]
            fixed_rows.add(r)
            fixed_cols.add(c)
            if _augment(r2, allowed, match, holder, fixed_cols):
                break
            fixed_rows.discard(r)
            fixed_cols.discard(c)
            match.clear()
            match.update(saved[0])
            holder.clear()
            holder.update(saved[1])
        fixed_rows.add(r)
        fixed_cols.add(match[r])
    return match


def _augment(r0, allowed, match, holder, fixed_cols):
    """An augmenting path for the unmatched row r0 over the columns not fixed (the rows that hold them are not fixed either);
    edits match / holder on success.  Iterative depth-first search: components of hundreds of rows (duplicated points) must
    not meet the interpreter's recursion limit."""
    seen = set()
    stack = [(r0, iter(sorted(allowed[r0])))]
    path = []                                             # (row, column) steps taken so far
    while stack:
        r, it = stack[-1]
        for c in it:
            if c in fixed_cols or c in seen:
                continue
            seen.add(c)
            r1 = holder.get(c)
            if r1 is None:                                # a free column: flip the path
                path.append((r, c))
                for rr, cc in path:
                    match[rr], holder[cc] = cc, rr
                return True
            path.append((r, c))
            stack.append((r1, iter(sorted(allowed[r1]))))
            break
        else:
            stack.pop()
            if path:
                path.pop()
    return False


def canonical_among_ties(cols, edges):
    """``cols`` [n]: an optimal assignment (row -> column); ``edges`` [E,2]: every (row, column) pair off it that is tight
    under its potentials (reart_lap_ties).  The optimal assignments are exactly the perfect matchings of the tight pairs,
    whatever optimal potentials they were drawn with; they differ from ``cols`` along alternating cycles, i.e. inside the
    strongly connected components of the row graph i -> owner(j).  Returns (the lexicographically smallest optimal assignment
    -- rows in ascending order, each the smallest column some optimum gives it --, number of rows that changed): a function of
    the SET of optima, so of the cost matrix alone."""
    cols = np.asarray(cols, dtype=np.int64).copy()
    n = cols.shape[0]
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if not len(e):
        return cols, 0
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    owner = np.empty(n, dtype=np.int64)
    owner[cols] = np.arange(n)
    to = owner[e[:, 1]]
    keep = to != e[:, 0]
    e, to = e[keep], to[keep]
    ncomp, lab = connected_components(coo_matrix((np.ones(len(e), np.int8), (e[:, 0], to)), shape=(n, n)), directed=True,
                                      connection="strong")
    inside = lab[e[:, 0]] == lab[to]                       # a pair between two components is in no perfect matching
    e = e[inside]
    if not len(e):
        return cols, 0
    changed = 0
    for comp in np.unique(lab[e[:, 0]]):
        rows = np.nonzero(lab == comp)[0]
        if len(rows) > MAX_TIED_ROWS:      # degenerate input (clouds of duplicated points: hundreds of rows tied with each other):
            canonical_among_ties.skipped += 1      # the solver's optimum stands for this component -- optimal, not canonical
            continue
        allowed = {int(r): {int(cols[r])} for r in rows}
        for i, j in e[lab[e[:, 0]] == comp]:
            allowed[int(i)].add(int(j))
        best = _lex_min_matching(rows, allowed, cols)
        for r, c in best.items():
            if cols[r] != c:
                cols[r] = c
                changed += 1
    assert np.array_equal(np.sort(cols), np.arange(n)), "canonical_among_ties: not a permutation"
    return cols, changed


# a tied component larger than this is left as the solver returned it (the host's exact choice is a matching problem per row of the
# component: fine for the 2-20 rows real ties involve, minutes for a cloud of exact duplicates)
MAX_TIED_ROWS = 256
canonical_among_ties.skipped = 0


def tight_pairs_host(src, tgt, cols, prices):
    """reart_lap_ties' pair list for ONE problem, computed on the host from reart_cdist's matrix (the kernel's overflow path:
    more tight pairs than it stores).  src, tgt [n,3] device tensors, cols [n], prices [n] -> [E,2] int64."""
    c = cdist(src[None], tgt[None])[0].cpu().numpy().astype(np.float64)
    p = prices.detach().cpu().numpy().astype(np.float64)
    col = np.asarray(cols.detach().cpu().numpy(), dtype=np.int64)
    both = torch.cat((src.reshape(-1), tgt.reshape(-1))).float()
    mx = 1.7320508 * float((both.max() - both.min()).item())           # fp32 difference, as the kernels take it
    if not mx > 0.0:
        mx = 1.0
    v = c + p[None, :]
    cur = v[np.arange(len(col)), col]
    hit = (v - cur[:, None]) <= mx * 1e-13
    hit[np.arange(len(col)), col] = False
    return np.argwhere(hit)


class TieBreaker:
    """reart_lap_ties behind a solve of B problems of n columns, and the host's canonical choice for the problems it flags.
    ``launch`` queues the two kernels and the copy of the B flags behind the solve (no synchronisation); after the caller's
    own synchronisation ``settle`` rewrites ``state["cols"]`` of every flagged problem IN PLACE and returns how many
    problems changed.  Buffers (caller-owned, as the C ABI wants them): per row ``K`` slots for the columns of its tight pairs
    and a count."""

    def __init__(self, B, n, device):
        self.B, self.n, self.device = B, n, device
        # slots per row.  Measured over the recipe's refreshes on nao (tools/exp_tie_counts.py): 46 % of the rows have no tight
        # pair off the assignment, 40 % one, 0.03 % more than eight -- but 27 % of the PROBLEMS hold such a row (the rows of a
        # region the searches flooded), 1 % one above 16, the largest 22.  A row beyond its slots sends its problem to the host
        # (flag 2), so the slots cover what occurs: 24 (the cycle check keeps n x K x 2 bytes in LDS: 8 above 2048 columns)
        self.K = 24 if n <= 2048 else 8
        self.tie = torch.zeros((B,), dtype=torch.int32, device=device)
        self.n_edges = torch.zeros((B, n), dtype=torch.int32, device=device)            # tight pairs per row
        self.edges = torch.zeros((B, n, self.K), dtype=torch.int32, device=device)      # their columns (the first K)
        self.tie_host = torch.zeros((B,), dtype=torch.int32).pin_memory()
        self._tie_np = self.tie_host.numpy()              # (the same memory: the per-solve question "any flag?" without a tensor op)
        self.flagged = self.changed = self.overflows = 0
        self.log = None                                   # set to [] to keep (problem, old cols, new cols, pairs) of every change

    def launch(self, src, tgt, cols, prices):
        _lib.check(_lib.lib().reart_lap_ties(_lib.ptr(src), _lib.ptr(tgt), self.B, self.n, _lib.ptr(cols), _lib.ptr(prices),
                                             _lib.ptr(self.tie), _lib.ptr(self.edges), _lib.ptr(self.n_edges), self.K, _lib.stream()),
                   "reart_lap_ties")
        self.tie_host.copy_(self.tie, non_blocking=True)

    def resolve_mc(self, src, tgt, racers, arr, cols, cert, prices, ws, copy=True):
        """reart_lap_resolve_points_mc_ties: the in-place re-solve with the tie check behind it -- the tight pairs come out of
        the solve's own certificate pass (no second scan of the costs); queues the copy of the flags like ``launch``
        (``copy=False``: the caller brings ``tie`` to the host itself and hands it over with ``flags_from``)."""
        _lib.check(_lib.lib().reart_lap_resolve_points_mc_ties(_lib.ptr(src), _lib.ptr(tgt), self.B, self.n, racers, arr, _lib.ptr(cols),
                                                               _lib.ptr(cert), _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(self.tie),
                                                               _lib.ptr(self.edges), _lib.ptr(self.n_edges), self.K, _lib.ptr(ws),
                                                               ws.numel(), _lib.stream()), "reart_lap_resolve_points_mc_ties")
        if copy:
            self.tie_host.copy_(self.tie, non_blocking=True)

    def flags_from(self, words):
        """The B tie flags as the caller read them (a numpy int32 view of its own pinned buffer)."""
        self._tie_np[:] = words

    def pairs_of(self, b):
        """[E,2] int64 (row, column) of problem b's tight pairs as the kernels listed them (rows with more than K: the first K)."""
        cnt = self.n_edges[b].cpu().numpy().clip(0, self.K)
        cols = self.edges[b].cpu().numpy()
        keep = np.arange(self.K)[None, :] < cnt[:, None]
        rows = np.broadcast_to(np.arange(self.n)[:, None], keep.shape)
        return np.stack((rows[keep], cols[keep]), axis=1).astype(np.int64)

    def settle(self, src, tgt, state, skip=()):
        """``skip``: problems whose solve did not certify (they went to the host solver) -- a collection, or a callable that
        returns one (only asked when some problem is flagged: the normal solve has none)."""
        if not self._tie_np.any():
            return 0
        if callable(skip):
            skip = skip()
        changed = 0
        cols, prices = state["cols"], state["prices"]
        for b in np.nonzero(self._tie_np)[0].tolist():
            if b in skip or int(self._tie_np[b]) == 3:
                continue
            self.flagged += 1
            if int(self._tie_np[b]) == 2:
                self.overflows += 1
                pairs = tight_pairs_host(src[b], tgt[b], cols[b], prices[b])
            else:
                pairs = self.pairs_of(b)
            old = cols[b].cpu().numpy()
            new, moved = canonical_among_ties(old, pairs)
            if moved:
                cols[b].copy_(torch.from_numpy(new).to(device=cols.device, dtype=cols.dtype))
                changed += 1
                if self.log is not None:
                    self.log.append((b, old.copy(), new.copy(), np.asarray(pairs).copy()))
        self.changed += changed
        return changed


def _tie_breaker(state, B, n, device):
    tb = state.get("tie_breaker")
    if tb is None or (tb.B, tb.n) != (B, n) or tb.device != device:
        tb = state["tie_breaker"] = TieBreaker(B, n, device)
    return tb


def canonicalize(src, tgt, state, skip=()):
    """One-shot form for the host-driven entries below: queue reart_lap_ties on ``state``'s optimum, wait, settle."""
    B, n = state["cols"].shape
    tb = _tie_breaker(state, B, n, src.device)
    cols = state["cols"] if state["cols"].dtype == torch.int32 else state["cols"].int()
    tb.launch(src, tgt, cols.contiguous(), state["prices"])
    torch.cuda.current_stream().synchronize()
    if cols is not state["cols"]:
        state["cols"] = cols
    return tb.settle(src, tgt, state, skip)


POINTS_NMAX = 2048   # reart_lap_resolve_points keeps both point sets and the solver state in LDS
# free-row orders raced per problem by linear_sum_assignment_points (reart_lap_resolve_points_race); REART_RESOLVE_RACERS=1: none
RESOLVE_RACERS = int(os.environ.get("REART_RESOLVE_RACERS", "13"))


# one search per wave (reart_lap_resolve_points_mw, csrc/lap_mw.hip) where its waves hold a whole problem in registers
# (Above 1024 columns -- 32 per lane -- the chains of ONE workgroup lose to the workgroup-wide row reduction: 24.5 against
# 20.4 ms per re-solve of the kinematic projection's 19 x 2048^2; spread over eight workgroups per problem they win: 16.6 ms.)
MW_NMIN, MW_NMAX = 512, int(os.environ.get("REART_RESOLVE_MW_NMAX", "2048"))
RESOLVE_PER_WAVE = os.environ.get("REART_RESOLVE_MW", "1") != "0"
# workgroups per problem of the row reduction on many compute units (reart_lap_resolve_points_mc); 0: the one-workgroup form
RESOLVE_ARR_WGS = int(os.environ.get("REART_RESOLVE_ARR_WGS", "-1"))     # -1: up to sixteen, as many as the batch leaves room for


# solver calls expected to be in flight at once on the device (the sweep's concurrent groups set it): the idle compute units a
# call hands to racers / reduction workgroups are shared among them
CONCURRENT_CALLS = 1


def _arr_wgs(B):
    # (x 8 chains each: with a team of four waves per chain -- lap_mc_arr_team_kernel -- 128 chains in flight per problem beat
    # 64, recipe refresh 2.68 -> 2.57 ms; 224 are no better)
    # (at least two: with the chip full of problems -- the recipe as a sweep, 4 x 95 in flight -- a problem's ~400 released rows on
    # the 8 chains of ONE workgroup are what a solve waits for: 20 x 15 000 with energies 26.0 s with 1, 24.3 with 2, 24.6 with 4)
    return RESOLVE_ARR_WGS if RESOLVE_ARR_WGS >= 0 else max(2, min(16, 512 // max(B * max(CONCURRENT_CALLS, 1), 1)))


def _resolve_racers(B, n, race=True):
    """Workgroups per problem of a re-solve: a racer holds a compute unit's LDS, the chip has 256; below 512 columns the
    re-solve is one short launch and is not raced."""
    if not race or n < 512:
        return 1
    r = min(RESOLVE_RACERS, 256 // max(B * max(CONCURRENT_CALLS, 1), 1))
    return r if r >= 2 else 1


def linear_sum_assignment_points(src, tgt, state, return_stats=False, race=True, per_wave=None, device_cols=False):
    """Optimal assignment for the Euclidean costs ``cdist(src, tgt)`` of two point batches [B,n,3] in a loop that
    re-solves slowly moving problems (the kinematic projection, run_robot.py:165-178 with ``--assign_gap 1``).  ``state`` is
    a dict the caller keeps between calls.  First call (or n > 2048): the cost matrices are built and solved like
    ``linear_sum_assignment_batch(cdist(src, tgt), state=state, warm_assignment=True)``.  Later calls with n <= 2048 never
    build a matrix: ``reart_lap_resolve_points`` re-solves from the previous optimum with the costs recomputed from the
    points inside the kernel (bit-equal to ``cdist``'s values), certificate included.  Same return value.
    ``race`` (default on): idle compute units run the same re-solve with the free rows taken in other orders and the first
    to finish publishes (``reart_lap_resolve_points_race``) -- the same optimum sooner; as with the cold race, the potentials
    kept in ``state`` are the winner's.
    ``per_wave`` (default: on for 512 <= n <= 2048): every wave of a problem's workgroup follows its own free row and commits
    under a lock (``reart_lap_resolve_points_mw``) instead of the whole workgroup following one row at a time: same optimum,
    timing-dependent potentials (like a race).  ``per_wave=("mc", W)``: the row reduction of every problem on W workgroups
    (``reart_lap_resolve_points_mc``).
    ``device_cols=True``: returns ``(cols, fallbacks)`` with ``cols`` the [B,n] int64 DEVICE tensor of assigned columns (rows
    are 0..n-1) instead of the host lists -- a loop that feeds the pairs back to the GPU (``RelaxEngine.set_assignment``) then
    only reads the B certificate flags on the host."""
    _lib.require_gpu(src, tgt)
    src, tgt = src.detach().contiguous().float(), tgt.detach().contiguous().float()
    B, n, _ = src.shape
    warm = (state.get("prices") is not None and state.get("cols") is not None and tuple(state["prices"].shape) == (B, n)
            and tuple(state["cols"].shape) == (B, n) and state["prices"].device == src.device)
    if not warm or n > POINTS_NMAX:
        res = linear_sum_assignment_batch(cdist(src, tgt), return_stats=(return_stats or device_cols), state=state,
                                          warm_assignment=True, points=(src, tgt), race=True)
        if CANONICAL_TIES and n <= 4096 and canonicalize(src, tgt, state) and not device_cols:
            rows_ = np.arange(n, dtype=np.int64)          # (a problem the host solved is optimal too: its ties are settled alike)
            fixed = [(rows_, c_) for c_ in state["cols"].cpu().numpy().astype(np.int64)]
            res = (fixed, *res[1:]) if isinstance(res, tuple) else fixed
        if device_cols:      # state["cols"] holds the certified (or host-solved) assignment
            return (state["cols"].long(), res[1], res[2]) if return_stats == "full" else (state["cols"].long(), res[1])
        return res
    L = _lib.lib()
    col, prices = state["cols"].clone(), state["prices"]
    cert = torch.zeros((B,), dtype=torch.int32, device=src.device)    # defined whatever the kernels write: 0 = solve on the host
    racers = _resolve_racers(B, n, race)
    if return_stats == "full":      # the statistics region of the workspace (shared with other calls): zeros unless written
        nb = L.reart_lap_race_workspace_bytes(B, n, racers) if racers > 1 else L.reart_lap_workspace_bytes(B, n)
        off = ((8 * B * n + 255) // 256) * 256
        _lib.workspace(nb, src.device)[off:off + 16 * B].zero_()
    if per_wave is None:
        per_wave = RESOLVE_PER_WAVE
    arr_wgs = int(per_wave[1]) if isinstance(per_wave, tuple) else (_arr_wgs(B) if per_wave else 0)
    # which form ran (statistics mean different things: the row-reduction steps of the chain forms are spread over many waves)
    state["resolve_form"] = ("mc" if arr_wgs > 0 else "mw") if (per_wave and MW_NMIN <= n <= MW_NMAX) else "jv"
    if per_wave and MW_NMIN <= n <= MW_NMAX and arr_wgs > 0:
        ws = _lib.workspace(L.reart_lap_mc_workspace_bytes(B, n, racers), src.device)
        rc = L.reart_lap_resolve_points_mc(_lib.ptr(src), _lib.ptr(tgt), B, n, racers, min(arr_wgs, 256), _lib.ptr(col), _lib.ptr(cert),
                                           _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), _lib.stream())
    elif per_wave and MW_NMIN <= n <= MW_NMAX:
        ws = _lib.workspace(L.reart_lap_race_workspace_bytes(B, n, racers), src.device)
        rc = L.reart_lap_resolve_points_mw(_lib.ptr(src), _lib.ptr(tgt), B, n, racers, _lib.ptr(col), _lib.ptr(cert),
                                           _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), _lib.stream())
    elif racers > 1:
        ws = _lib.workspace(L.reart_lap_race_workspace_bytes(B, n, racers), src.device)
        rc = L.reart_lap_resolve_points_race(_lib.ptr(src), _lib.ptr(tgt), B, n, racers, _lib.ptr(col), _lib.ptr(cert),
                                             _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), _lib.stream())
    else:
        ws = _lib.workspace(L.reart_lap_workspace_bytes(B, n), src.device)
        rc = L.reart_lap_resolve_points(_lib.ptr(src), _lib.ptr(tgt), B, n, _lib.ptr(col), _lib.ptr(cert), _lib.ptr(prices),
                                        _lib.ptr(prices), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_lap_resolve_points")
    state["cols"] = col
    tb = None
    if CANONICAL_TIES:
        tb = _tie_breaker(state, B, n, src.device)
        tb.launch(src, tgt, col, prices)
    cert_h = cert.cpu().numpy()
    if tb is not None:
        tb.settle(src, tgt, state, skip=set(np.nonzero(cert_h == 0)[0].tolist()))
    col_h = None if device_cols else col.cpu().numpy().astype(np.int64)
    rows = np.arange(n, dtype=np.int64)
    out, fallbacks = [], 0
    for b in range(B):
        if cert_h[b]:
            if not device_cols:
                out.append((rows, col_h[b]))
        else:  # certificate did not close: exact host solve for this matrix
            from scipy.optimize import linear_sum_assignment

            fallbacks += 1
            out.append(linear_sum_assignment(cdist(src[b:b + 1], tgt[b:b + 1])[0].cpu().numpy()))
            _forget_uncertified(state, state["cols"], b, out[-1][1])
    if device_cols and return_stats != "full":
        return state["cols"].long(), fallbacks
    if return_stats == "full":
        off = ((8 * B * n + 255) // 256) * 256
        st = ws[off:off + 16 * B].view(torch.int32).reshape(B, 4).cpu().numpy().copy()
        if per_wave and MW_NMIN <= n <= MW_NMAX:      # the per-wave row reduction reports its redone steps in the upper half
            state["commit_conflicts"] = (st[:, 1] >> 16) & 0xffff
            st[:, 1] &= 0xffff
            # ... and the rounds of the backward growth (lap_mc_forest_kernel: sequential workgroup-wide steps like the searches')
            state["backward_rounds"] = (st[:, 0] >> 21) & 0x3ff
            st[:, 0] &= 0x1fffff
            state["winner"] = (st[:, 0] >> 16) & 31          # the racer that published (bits 16-20 of word 0; released rows below)
        return (state["cols"].long() if device_cols else out), fallbacks, st
    return (out, fallbacks) if return_stats else out


def spatial_order(pts):
    """[B,n,3] points -> [B,n] long: a permutation per problem that numbers them along a Z-order (Morton) curve.
    The loops that re-solve against FIXED targets (run_robot.py:167-169 samples them once) number their columns this way: the
    searches' workgroups deal column j to thread j mod (workgroup size), so the 64 columns of a wave become neighbours in
    space, and the few columns an entry of a bucket round improves -- neighbours of its row -- sit in one or two waves instead
    of in most of them (the exact path of the relaxations is executed per wave).  On replayed solves (tools/replay_tail.py
    ORDER=morton): recipe 219 -> 199 ms over the slowest 24, projection 467 -> 400.  Any permutation gives the same optimum."""
    q = pts.float() - pts.float().amin(dim=1, keepdim=True)
    q = (q / q.amax(dim=(1, 2), keepdim=True).clamp_min(1e-30) * 1023.0).long().clamp_(0, 1023)
    code = torch.zeros(pts.shape[:2], dtype=torch.long, device=pts.device)
    for bit in range(10):
        for ax in range(3):
            code |= ((q[..., ax] >> bit) & 1) << (3 * bit + ax)
    return torch.argsort(code, dim=1, stable=True)


class ReplayedLaunches:
    """The launches of one refresh, queued eagerly until the same ``key`` (every address and launch parameter the closure
    uses) has come ``after`` times in a row, then captured ONCE into a graph and replayed.  A refresh is ~18 launches (passes,
    set-up, tighten, row reduction, trees, backward growth, searches, certificate, tie check, the copies of the flags), each too
    short to hide the ~15 us between two eagerly queued kernels: a quarter of a millisecond of a 1.3 ms solve
    (profiles/r06_solve_spans_nao_projection.txt).  Same kernels, same arguments.  REART_RESOLVE_GRAPH=0: always eager.
    ``run(key, fn)`` queues or replays; after the caller's own synchronisation ``settle(fn, ok)`` captures when due -- a
    capture queues nothing, so the state is as the eager refresh left it (``ok`` False -- a host fallback touched the
    state -- puts the capture off).  ``guard``: a context-manager factory entered around the capture (the sweep's gate)."""

    ENABLED = os.environ.get("REART_RESOLVE_GRAPH", "1") != "0"

    def __init__(self, after=2):
        self.after, self.key, self.graph, self.seen, self.replays, self.guard = after, None, None, 0, 0, None

    def run(self, key, fn):
        if key != self.key:
            self.key, self.graph, self.seen = key, None, 0
        if self.graph is not None:
            self.graph.replay()
            self.replays += 1
        else:
            fn()
            self.seen += 1

    def settle(self, fn, ok=True):
        if not self.ENABLED or self.graph is not None or self.seen < self.after or not ok:
            return
        with (self.guard() if self.guard is not None else contextlib.nullcontext()):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
        self.graph = g


class InPlaceResolve:
    """The re-solve of a loop that keeps its problems on the device (run_robot.py:164-187 with the pairs fed straight back to
    the GPU): ``state["cols"]`` (int32 [B,n]) and ``state["prices"]`` (f64 [B,n]) of an earlier solve are re-solved IN PLACE
    for new source points (``reart_lap_resolve_points_mc``), so that launches which read them -- a captured graph included --
    see the new optimum at the same addresses.  The host waits for the B certificate flags (and, on request, the B x 4
    statistics words) and nothing else; an uncertified problem goes to scipy like everywhere.  ``usable(state, B, n)`` says
    whether the state qualifies (a first call has to go through ``linear_sum_assignment_points``)."""

    def __init__(self, B, n, device):
        self.B, self.n, self.device = B, n, device
        self.cert = torch.zeros((B,), dtype=torch.int32, device=device)
        # what the host reads after a refresh, in ONE pinned buffer that the refresh's last launch writes (reart_publish_words):
        # certificate flags | tie flags | the statistics words
        self.words_host = torch.zeros((6 * B,), dtype=torch.int32).pin_memory()
        self._words_np = self.words_host.numpy()
        self.cert_host, self.stats_host = self.words_host[:B], self.words_host[2 * B:]
        self._ws = None                      # owned: a captured graph keeps its address
        self.launches = ReplayedLaunches()
        self._event = self._pending = None

    @staticmethod
    def usable(state, B, n):
        return (RESOLVE_PER_WAVE and MW_NMIN <= n <= MW_NMAX and _arr_wgs(B) > 0 and state.get("cols") is not None
                and state.get("prices") is not None and state["cols"].dtype == torch.int32 and tuple(state["cols"].shape) == (B, n)
                and tuple(state["prices"].shape) == (B, n))

    def _queue(self, src, tgt, cols, prices, racers, arr, stats, tb):
        """Every launch and copy of one refresh on the current stream (no synchronisation)."""
        L, B, n = _lib.lib(), self.B, self.n
        ws = self._ws
        off = ((8 * B * n + 255) // 256) * 256                            # the solver's statistics: [B][4] ints behind the potentials
        # (the flags and the statistics start defined inside the call: its set-up launch clears them)
        if tb is not None:
            tb.resolve_mc(src, tgt, racers, arr, cols, self.cert, prices, ws, copy=False)
        else:
            _lib.check(L.reart_lap_resolve_points_mc(_lib.ptr(src), _lib.ptr(tgt), B, n, racers, arr, _lib.ptr(cols), _lib.ptr(self.cert),
                                                     _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), _lib.stream()),
                       "reart_lap_resolve_points_mc")
        # (without a tie check the middle words repeat the certificate flags: the layout stays put)
        _lib.check(L.reart_publish_words(_lib.ptr(self.cert), B, _lib.ptr(tb.tie if tb is not None else self.cert), B,
                                         _lib.c_void_p(ws.data_ptr() + off) if stats else None, 4 * B if stats else 0,
                                         _lib.c_void_p(self.words_host.data_ptr()), _lib.stream()), "reart_publish_words")

    def begin(self, src, tgt, state, stats=False):
        """Queue the refresh on the current stream and record an event behind it; nothing waits.  The caller may queue more
        work that READS ``state["cols"]`` behind it before ``finish`` -- and must queue that work again if ``finish`` says the
        columns changed on the host (a tied or an uncertified problem: about one refresh in a thousand)."""
        L, B, n = _lib.lib(), self.B, self.n
        racers, arr = _resolve_racers(B, n), min(_arr_wgs(B), 256)
        need = L.reart_lap_mc_workspace_bytes(B, n, racers)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        cols, prices = state["cols"], state["prices"]
        tb = _tie_breaker(state, B, n, self.device) if CANONICAL_TIES else None      # --deterministic: the tied problems, with the solve
        key = (src.data_ptr(), tgt.data_ptr(), cols.data_ptr(), prices.data_ptr(), self._ws.data_ptr(), racers, arr, bool(stats), id(tb))
        queue = lambda: self._queue(src, tgt, cols, prices, racers, arr, stats, tb)
        self.launches.run(key, queue)
        state["resolve_form"] = "mc"
        if self._event is None:
            self._event = torch.cuda.Event()
        self._event.record()
        self._pending = (src, tgt, state, stats, tb, queue)

    def finish(self):
        """Wait for the refresh queued by ``begin`` (its event, not the stream) and settle it on the host.
        -> (host fallbacks, [B,4] int32 numpy statistics as the kernel wrote them or None, columns changed on the host)"""
        (src, tgt, state, stats, tb, queue), self._pending = self._pending, None
        B = self.B
        self._event.synchronize()
        fb, changed = 0, 0
        words = self._words_np
        bad = np.nonzero(words[:B] == 0)[0].tolist()
        if tb is not None:
            tb.flags_from(words[B:2 * B])
            changed = tb.settle(src, tgt, state, skip=bad)
        raw = words[2 * B:].reshape(B, 4).copy() if stats else None
        self.launches.settle(queue, ok=not bad)
        for b in bad:                                                     # certificate did not close: exact host solve
            from scipy.optimize import linear_sum_assignment

            fb += 1
            host = linear_sum_assignment(cdist(src[b:b + 1], tgt[b:b + 1])[0].cpu().numpy())[1]
            _forget_uncertified(state, state["cols"], b, host)
        return fb, raw, bool(changed or fb)

    def __call__(self, src, tgt, state, stats=False):
        """-> (host fallbacks, [B,4] int32 numpy statistics as the kernel wrote them or None)"""
        self.begin(src, tgt, state, stats)
        fb, raw, _ = self.finish()
        return fb, raw
