#!/usr/bin/env python3
"""Drop-in counterpart of the reference's ``run_robot.py`` optimisation loop (``:154-221``) on the
HIP path: same command-line flags and defaults (``run_robot.py:362-420``), same checkpoint keys
(``:340-356``), same loss branches.

What is built: the loop itself.
  * ``--model base`` without assignment loss (the first ``--assign_iter`` iterations of every run,
    and whole runs without ``--use_assign_loss``): the fused ``RelaxEngine`` -- five kernel launches
    per iteration replayed from a graph, no host sync inside the loop.
  * ``--model base`` with ``--use_assign_loss`` after ``--assign_iter``: the same engine in its
    assignment-loss mode; every ``--assign_gap`` iterations the host refreshes the pairs (forward of the
    coming iteration, FPS, cost matrices, GPU linear assignment), in between the engine runs alone.
  * ``--model kinematic``: the reference's own loop structure with
    the HIP operators underneath (``BaseModel`` / ``KinematicModel``, ``ChamferDistance``,
    ``blend_anchor_motion``, ``flow_loss``, FPS) and ``torch.optim.Adam``; the linear assignment of the
    reference (``run_robot.py:172-176``, scipy on a process pool) runs on the GPU (``reart_lap_auction``:
    auction + exact dual certificate, host fallback for an uncertified matrix).
  * end of run (``run_robot.py:224-330``): ``reart_amd.tail`` -- denoise / merge / spanning tree / renumbering,
    the model-selection energy (assignment, screw and group errors), ``result.pkl`` / ``result.txt`` /
    ``model.pth.tar`` with the reference's keys; ``--model kinematic --base_result_path result.pkl`` builds the
    joint tree from a base result (``run_robot.py:101-124``).
What is NOT built (SURVEY.md section 8 out of scope): visualisation (gif / html) and the GT-graph tree edit distance
(``apted``); ground-truth metrics and the retargeting error (``ik``) are reported when the sequence carries them.

Data: ``--seq_path`` with the reference's pickle layout (``reart_amd.dataset.Sequence`` = ``dataset/dataset_robot.py``), or
``--synthetic`` for the generated articulated sequence of ``reart_amd/synthetic.py``.
"""
import argparse
import contextlib
import ctypes
import functools
import glob
import os
import pickle
import random

import numpy as np
import torch

from reart_amd.knn_cuda import KNN
from reart_amd.networks.loss import flow_loss, recon_loss
from reart_amd.networks.model import BaseModel, KinematicModel
from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points
from reart_amd.relax import RelaxEngine
from reart_amd.utils.chamfer import ChamferDistance
from reart_amd.utils.flow_utils import blend_anchor_motion
from reart_amd.utils.model_utils import tau_cosine


def load_sequence(seq_path, num_points, cano_idx):
    """The reference's sample dictionary (dataset/dataset_robot.py:9-100) through the mirror loader."""
    from reart_amd.dataset import Sequence

    return Sequence(seq_path, num_points=num_points, cano_idx=cano_idx)[0]


def synthetic_sequence(num_points, cano_idx, frames, with_flow, seed=2):
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=frames, n_parts=8, pts_per_part=num_points // 8, seed=seed, with_flow=with_flow)
    cano, pc_list = split_canonical(seq["complete"], cano_idx)
    out = dict(cano_pc=cano, pc_list=pc_list, complete_pc_list=seq["complete"], gt_cano_part=seq["part"])
    if with_flow:
        out.update(ref_loc=seq["ref_loc"], ref_flow=seq["ref_flow"])
    return out


def flow_references(args, sample, device, seq_path=None):
    """The flow branch's reference sets (run_robot.py:64-84) -> (pc_ref_list, flow_ref_list), lists of [M_i,3] tensors:
    the generator's own references for a synthetic sample, otherwise descriptors -> SMNN matches -> reference flows."""
    if "ref_loc" in sample:  # synthetic references (true motion + noise)
        return ([torch.from_numpy(r).to(device) for r in sample["ref_loc"]],
                [torch.from_numpy(f).to(device) for f in sample["ref_flow"]])
    from reart_amd.networks.feature_extractor import get_extractor
    from reart_amd.utils.dataset_utils import load_normalize_dict
    from reart_amd.utils.flow_utils import compute_corr_list_filter, normalize_pc_list

    extractor = get_extractor(args)
    info = load_normalize_dict(args.normalize_file)[os.path.basename(seq_path.rstrip("/"))]
    centroid = torch.from_numpy(info["centroid"]).float().to(device)
    complete = torch.from_numpy(sample["complete_pc_list"]).float().to(device)
    norm = normalize_pc_list(complete, centroid, info["scale"].item())
    src_list, tgt_list = compute_corr_list_filter(norm, extractor, None, matching="smnn")
    return ([complete[i][s] for i, s in enumerate(src_list)],
            [complete[i + 1][t] - complete[i][s] for i, (s, t) in enumerate(zip(src_list, tgt_list))])


class AssignmentPhase:
    """The base model's assignment-loss phase (run_robot.py:164-187) on the fused ``RelaxEngine``.  Every ``assign_gap``
    iterations: forward of the coming iteration (``peek_forward``: the temperature and noise the step will use) -> the sampled
    source points (FPS of the canonical cloud, run_robot.py:167) -> optimal assignment to the sampled target points
    (``cdist`` + ``linear_sum_assignment`` in the reference, :170-176) -> the pairs go to the engine, which then runs the
    iterations up to the next refresh from a captured graph without touching the host.
    Both FPS calls of the reference sample fixed clouds from a fixed start (its CUDA FPS starts at index 0): computed once.
    The first refresh is a cold solve (raced auction + exact certificate); every later one re-solves from the previous
    optimum and potentials (``linear_sum_assignment_points``: shortest augmenting paths, the row reduction one chain per
    wave, costs recomputed from the points, same certificate).  ``fallbacks`` counts matrices that went to the host solver."""

    def __init__(self, eng, cano_pc, pc_list, downsample, assign_gap, lambda_assign):
        self.eng, self.gap, self.lam = eng, int(assign_gap), float(lambda_assign)
        device = cano_pc.device
        self.B, N = pc_list.shape[0], pc_list.shape[1]
        self.n = N // downsample
        zero = torch.zeros(1, dtype=torch.long, device=device)
        self.src_idx = farthest_point_sample(cano_pc[None], self.n, start=zero, cuda_mode=True)                      # [1, n]
        tgt_fps = farthest_point_sample(pc_list, self.n, start=zero.expand(self.B), cuda_mode=True)                  # [B, n]
        # the sampled targets are the COLUMNS of every refresh: numbered along a Z-order curve (lap.spatial_order: the searches'
        # waves then hold neighbouring columns); the pairs a refresh returns are the same whatever the numbering
        from reart_amd.utils.lap import spatial_order
        self.tgt_order = spatial_order(index_points(pc_list, tgt_fps))                                               # [B, n] into the FPS order
        self.tgt_idx = tgt_fps.gather(1, self.tgt_order)
        self.tgt_pts = index_points(pc_list, self.tgt_idx).contiguous()
        self.lap_state = {}
        self.refreshes = self.fallbacks = 0
        self.events = None          # set to [] to collect a (start, end) torch.cuda.Event pair around every solve
        self.collect_stats = False  # True: the solver's per-problem statistics of every refresh are read (B x 4 ints) into stats_log
        self.stats_log = []         # per refresh: (sequential steps of the slowest problem = search steps + backward rounds, mean search
                                    # steps, mean row-reduction steps, most search steps, mean backward rounds)
        self.stats_raw = []         # per device-side refresh: the solver's [B,4] statistics words as written (tools/exp_tail.py)
        self.capture_guard = None   # a context-manager factory entered around the graph capture (the sweep's _CaptureGate)
        self._have = False

    def _native_tables(self):
        """Index tables of the device-side refresh, in the engine's STORAGE order (built once): the stored position of every
        sampled canonical point, the sample slot of every stored point (-1: not sampled), the stored position of every
        sampled target point."""
        eng, dev = self.eng, self.eng.device
        N = eng._assign_map.shape[1]
        src = self.src_idx[0].long()
        tgt = self.tgt_idx.long()
        if eng._perm is not None:
            src = eng._inv[src]
            if getattr(eng, "_inv_frames", None) is None:
                inv_f = torch.empty_like(eng._perm_frames)
                inv_f.scatter_(1, eng._perm_frames, torch.arange(N, device=dev).expand(self.B, N))
                eng._inv_frames = inv_f
            tgt = eng._inv_frames.gather(1, tgt)
        slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
        slot[src] = torch.arange(self.n, dtype=torch.int32, device=dev)
        self._src_stored, self._slot, self._tgt_stored = src.int().contiguous(), slot, tgt.int().contiguous()
        self._src_pts = torch.empty((self.B, self.n, 3), device=dev)
        self._cert = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        # what the host reads after a refresh -- certificate flags | tie flags | statistics -- in one pinned buffer, written by the
        # refresh's last launch (reart_publish_words)
        self._words_host = torch.zeros((6 * self.B,), dtype=torch.int32).pin_memory()
        self._words_np = self._words_host.numpy()
        self._cert_host, self._stats_host = self._words_host[:self.B], self._words_host[2 * self.B:]

    def _refresh_on_device(self):
        """A refresh after the first, without a host-side tensor operation: gather (reart_gather_points) -> re-solve from the
        previous optimum in place (reart_lap_resolve_points_mc) -> pair map (reart_assign_pairs), all queued behind the
        forward; the host waits for the B certificate flags only (an uncertified problem goes to scipy, as everywhere)."""
        from reart_amd import _lib
        from reart_amd.utils import lap

        eng, st, L = self.eng, self.lap_state, _lib.lib()
        B, n, N = self.B, self.n, eng._assign_map.shape[1]
        racers, arr = lap._resolve_racers(B, n), min(lap._arr_wgs(B), 256)
        need = L.reart_lap_mc_workspace_bytes(B, n, racers)
        if getattr(self, "_ws", None) is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=eng.device)       # owned: a captured graph keeps its address
            self._launches = lap.ReplayedLaunches()
        ws, stats = self._ws, bool(self.collect_stats)
        self._launches.guard = self.capture_guard
        cols, prices = st["cols"], st["prices"]
        off = ((8 * B * n + 255) // 256) * 256                            # the solver's statistics: [B][4] ints behind the potentials
        tb = lap._tie_breaker(st, B, n, eng.device) if lap.CANONICAL_TIES else None      # --deterministic: tied optima take the canonical one

        def pairs():
            _lib.check(L.reart_assign_pairs(_lib.ptr(cols), _lib.ptr(self._slot), _lib.ptr(self._tgt_stored), B, N, n,
                                            _lib.ptr(eng._assign_map), _lib.stream()), "reart_assign_pairs")

        def queue():                                                      # every launch and copy of the refresh (lap.ReplayedLaunches)
            stream = _lib.stream()
            _lib.check(L.reart_gather_points(_lib.ptr(eng._pc_trans), _lib.ptr(self._src_stored), B, N, n, _lib.ptr(self._src_pts), stream),
                       "reart_gather_points")
            if tb is not None:                                            # (flags and statistics are cleared by the call's set-up launch)
                tb.resolve_mc(self._src_pts, self.tgt_pts, racers, arr, cols, self._cert, prices, ws, copy=False)
            else:
                _lib.check(L.reart_lap_resolve_points_mc(_lib.ptr(self._src_pts), _lib.ptr(self.tgt_pts), B, n, racers, arr, _lib.ptr(cols),
                                                         _lib.ptr(self._cert), _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), stream),
                           "reart_lap_resolve_points_mc")
            pairs()
            _lib.check(L.reart_publish_words(_lib.ptr(self._cert), B, _lib.ptr(tb.tie if tb is not None else self._cert), B,
                                             _lib.c_void_p(ws.data_ptr() + off) if stats else None, 4 * B if stats else 0,
                                             _lib.c_void_p(self._words_host.data_ptr()), stream), "reart_publish_words")

        self._launches.run((eng._pc_trans.data_ptr(), eng._assign_map.data_ptr(), cols.data_ptr(), prices.data_ptr(), racers, arr, stats, id(tb)), queue)
        st["resolve_form"] = "mc"
        torch.cuda.current_stream().synchronize()
        bad = np.nonzero(self._words_np[:B] == 0)[0].tolist()
        if tb is not None:
            tb.flags_from(self._words_np[B:2 * B])
            if tb.settle(self._src_pts, self.tgt_pts, st, skip=bad):
                pairs()
        self._launches.settle(queue, ok=not bad)
        if self.collect_stats:
            sth = self._stats_host.numpy().reshape(B, 4)
            self.stats_raw.append(sth.copy())
            back = (sth[:, 0] >> 21) & 0x3ff               # rounds of the backward growth: sequential steps of the same kind
            self.stats_log.append((int((sth[:, 2] + back).max()), float(sth[:, 2].mean()), float((sth[:, 3] >> 8).mean()),
                                   int(sth[:, 2].max()), float(back.mean())))
        fb = 0
        for b in bad:                                                      # certificate did not close: exact host solve
            from scipy.optimize import linear_sum_assignment

            fb += 1
            host = linear_sum_assignment(lap.cdist(self._src_pts[b:b + 1], self.tgt_pts[b:b + 1])[0].cpu().numpy())[1]
            lap._forget_uncertified(st, cols, b, host)
        if fb:
            pairs()
        return fb

    def _device_path(self):
        from reart_amd.utils import lap

        st = self.lap_state
        return (self.NATIVE and self.eng.cfg.use_assign and self.eng.cfg.lambda_assign == ctypes.c_float(self.lam).value
                and st.get("cols") is not None and st.get("prices") is not None and st["cols"].dtype == torch.int32
                and tuple(st["cols"].shape) == (self.B, self.n) and lap.RESOLVE_PER_WAVE and lap.MW_NMIN <= self.n <= lap.MW_NMAX
                and lap._arr_wgs(self.B) > 0)

    NATIVE = os.environ.get("REART_ASSIGN_NATIVE", "1") != "0"

    def refresh(self):
        from reart_amd.utils.lap import linear_sum_assignment_points

        eng = self.eng
        eng.peek_forward()
        if self._device_path():
            if getattr(self, "_slot", None) is None:
                self._native_tables()
            if self.events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            fb = self._refresh_on_device()
            if self.events is not None:
                ev[1].record()
                self.events.append(ev)
            self.fallbacks += fb
            self.refreshes += 1
            return False
        src_pts = index_points(eng.pc_trans, self.src_idx.expand(self.B, self.n)).contiguous()
        if self.events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.collect_stats:
            cols, fb, st = linear_sum_assignment_points(src_pts, self.tgt_pts, self.lap_state, device_cols=True, return_stats="full")
            back = np.asarray(self.lap_state.get("backward_rounds", np.zeros(len(st), np.int64)), dtype=np.int64)
            self.stats_log.append((int((st[:, 2] + back).max()), float(st[:, 2].mean()), float((st[:, 3] >> 8).mean()),
                                   int(st[:, 2].max()), float(back.mean())))
        else:
            cols, fb = linear_sum_assignment_points(src_pts, self.tgt_pts, self.lap_state, device_cols=True)
        if self.events is not None:
            ev[1].record()
            self.events.append(ev)
        self.fallbacks += fb
        self.refreshes += 1
        first = not eng.cfg.use_assign
        eng.set_assignment(self.src_idx[0], self.tgt_idx.gather(1, cols), self.lam)
        self._have = True
        return first

    def run(self, i, n_iter, snapshot_gap=0, on_snapshot=None):
        """Iterations i .. n_iter-1 -> n_iter.  ``on_snapshot(k)`` after iteration k-1 whenever k % snapshot_gap == 0 or k == n_iter."""
        eng = self.eng
        while i < n_iter:
            if not self._have or i % self.gap == 0:
                if self.refresh() and n_iter - i > 1:
                    # the engine's mode changed (its graph was dropped): one eager iteration, then a graph of the iterations
                    # between two refreshes
                    if self.gap > 1:
                        with (self.capture_guard() if self.capture_guard else contextlib.nullcontext()):
                            i += eng.capture(steps_per_graph=self.gap - 1)
                        # the capture's eager iteration has moved i: a refresh or a snapshot may be due right here
                        # (assign_iter % assign_gap == assign_gap - 1: run_robot.py:165 refreshes at every i % gap == 0)
                        if on_snapshot is not None and (i == n_iter or (snapshot_gap and i % snapshot_gap == 0)):
                            on_snapshot(i)
                        continue
            nxt = (i // self.gap + 1) * self.gap                                       # next refresh
            snap = (i // snapshot_gap + 1) * snapshot_gap if snapshot_gap else n_iter
            chunk = max(1, min(nxt, snap, n_iter) - i)
            eng.step(chunk)
            i += chunk
            if on_snapshot is not None and (i == n_iter or (snapshot_gap and i % snapshot_gap == 0)):
                on_snapshot(i)
        return i

    def report(self):
        out = {"assign_refreshes": self.refreshes, "lap_fallbacks": self.fallbacks}
        if self.events:
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self.events]
            out.update(ms_per_solve=sum(ms[1:]) / max(len(ms) - 1, 1), first_solve_ms=ms[0])
        return out


class AssignmentPhaseBatch:
    """The assignment phase of K instances of one shape that step in shared launches (``RelaxBatch``; the sweep over canonical
    frames): every refresh gathers the sampled source points of all instances and solves their K x (T-1) assignment problems
    in ONE call -- one set of launches, every problem its own workgroup(s) --, then hands each instance its pairs; the
    iterations between refreshes replay one graph for all instances.  Each instance ends exactly as its own
    ``AssignmentPhase`` would (the optimal assignment does not depend on who else is in the call)."""

    def __init__(self, batch, clouds, downsample, assign_gap, lambda_assign):
        """clouds: [(cano_pc, pc_list)] per engine of ``batch``, in the caller's point order."""
        self.batch, self.gap, self.lam = batch, int(assign_gap), float(lambda_assign)
        self.parts = [AssignmentPhase(e, c, p, downsample, assign_gap, lambda_assign) for e, (c, p) in zip(batch.engines, clouds)]
        if len({(ph.B, ph.n) for ph in self.parts}) != 1:
            raise ValueError("AssignmentPhaseBatch: the instances must share one shape")
        self.B, self.n = self.parts[0].B, self.parts[0].n
        self.tgt_all = torch.cat([ph.tgt_pts for ph in self.parts], dim=0).contiguous()
        self.lap_state = {}
        self.refreshes = self.fallbacks = 0
        self.capture_guard = None
        self.work_guard = None      # a context-manager factory entered around every refresh (the sweep's gate, reader side)
        self._have = False

    def refresh(self):
        if self.work_guard is None:
            first = self._refresh()
        else:
            with self.work_guard():
                first = self._refresh()
        due, self._capture_due = getattr(self, "_capture_due", None), None
        if due is not None:                     # the refresh's own graph (lap.ReplayedLaunches): captured OUTSIDE the reader section
            self._launches.settle(*due)         # of the gate -- a capture is its writer
        return first

    def _refresh(self):
        from reart_amd.utils.lap import linear_sum_assignment_points

        for ph in self.parts:
            ph.eng.peek_forward()
        if self._device_path():
            self.fallbacks += self._refresh_on_device()
            self.refreshes += 1
            return False
        src_all = torch.cat([index_points(ph.eng.pc_trans, ph.src_idx.expand(self.B, self.n)) for ph in self.parts], dim=0).contiguous()
        cols, fb = linear_sum_assignment_points(src_all, self.tgt_all, self.lap_state, device_cols=True)
        self.fallbacks += fb
        self.refreshes += 1
        first = not self.parts[0].eng.cfg.use_assign
        for k, ph in enumerate(self.parts):
            ph.eng.set_assignment(ph.src_idx[0], ph.tgt_idx.gather(1, cols[k * self.B:(k + 1) * self.B]), self.lam)
        self._have = True
        return first

    def _device_path(self):
        from reart_amd.utils import lap

        st, K = self.lap_state, len(self.parts)
        return (AssignmentPhase.NATIVE and all(ph.eng.cfg.use_assign and ph.eng.cfg.lambda_assign == ctypes.c_float(self.lam).value for ph in self.parts)
                and st.get("cols") is not None and st.get("prices") is not None and st["cols"].dtype == torch.int32
                and tuple(st["cols"].shape) == (K * self.B, self.n) and lap.RESOLVE_PER_WAVE
                and lap.MW_NMIN <= self.n <= lap.MW_NMAX and lap._arr_wgs(K * self.B) > 0)

    def _refresh_on_device(self):
        """AssignmentPhase._refresh_on_device for the K instances of a shared launch: K gathers into one source batch, ONE
        re-solve of the K x (T-1) problems, K pair maps; the host waits for the certificate flags only."""
        from reart_amd import _lib
        from reart_amd.utils import lap

        st, L, K, B, n = self.lap_state, _lib.lib(), len(self.parts), self.B, self.n
        dev = self.parts[0].eng.device
        if getattr(self, "_src_all", None) is None:
            for ph in self.parts:
                ph._native_tables()
            self._src_all = torch.empty((K * B, n, 3), device=dev)
            self._cert = torch.zeros((K * B,), dtype=torch.int32, device=dev)
            self._words_host = torch.zeros((2 * K * B,), dtype=torch.int32).pin_memory()       # certificate flags | tie flags (reart_publish_words)
            self._words_np = self._words_host.numpy()
            self._cert_host = self._words_host[:K * B]
        racers, arr = lap._resolve_racers(K * B, n), min(lap._arr_wgs(K * B), 256)
        need = L.reart_lap_mc_workspace_bytes(K * B, n, racers)
        if getattr(self, "_ws", None) is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)              # owned: a captured graph keeps its address
            self._launches = lap.ReplayedLaunches()
        ws = self._ws
        self._launches.guard = self.capture_guard
        cols, prices = st["cols"], st["prices"]
        tb = lap._tie_breaker(st, K * B, n, dev) if lap.CANONICAL_TIES else None

        def pairs():
            for k, ph in enumerate(self.parts):
                N = ph.eng._assign_map.shape[1]
                _lib.check(L.reart_assign_pairs(_lib.ptr(cols[k * B:]), _lib.ptr(ph._slot), _lib.ptr(ph._tgt_stored), B, N, n,
                                                _lib.ptr(ph.eng._assign_map), _lib.stream()), "reart_assign_pairs")

        def queue():                                                      # every launch and copy of the refresh (lap.ReplayedLaunches)
            stream = _lib.stream()
            for k, ph in enumerate(self.parts):
                N = ph.eng._assign_map.shape[1]
                _lib.check(L.reart_gather_points(_lib.ptr(ph.eng._pc_trans), _lib.ptr(ph._src_stored), B, N, n,
                                                 _lib.ptr(self._src_all[k * B:]), stream), "reart_gather_points")
            if tb is not None:                                            # (flags are cleared by the call's set-up launch)
                tb.resolve_mc(self._src_all, self.tgt_all, racers, arr, cols, self._cert, prices, ws, copy=False)
            else:
                _lib.check(L.reart_lap_resolve_points_mc(_lib.ptr(self._src_all), _lib.ptr(self.tgt_all), K * B, n, racers, arr, _lib.ptr(cols),
                                                         _lib.ptr(self._cert), _lib.ptr(prices), _lib.ptr(prices), _lib.ptr(ws), ws.numel(), stream),
                           "reart_lap_resolve_points_mc")
            pairs()
            _lib.check(L.reart_publish_words(_lib.ptr(self._cert), K * B, _lib.ptr(tb.tie if tb is not None else self._cert), K * B, None, 0,
                                             _lib.c_void_p(self._words_host.data_ptr()), stream), "reart_publish_words")

        key = tuple(ph.eng._pc_trans.data_ptr() for ph in self.parts) + tuple(ph.eng._assign_map.data_ptr() for ph in self.parts)
        self._launches.run(key + (cols.data_ptr(), prices.data_ptr(), racers, arr, id(tb)), queue)
        st["resolve_form"] = "mc"
        torch.cuda.current_stream().synchronize()
        bad = np.nonzero(self._words_np[:K * B] == 0)[0].tolist()
        if tb is not None:
            tb.flags_from(self._words_np[K * B:])
            if tb.settle(self._src_all, self.tgt_all, st, skip=bad):
                pairs()
        self._capture_due = (queue, not bad)
        fb = 0
        for b in bad:
            from scipy.optimize import linear_sum_assignment

            fb += 1
            host = linear_sum_assignment(lap.cdist(self._src_all[b:b + 1], self.tgt_all[b:b + 1])[0].cpu().numpy())[1]
            lap._forget_uncertified(st, cols, b, host)
        if fb:
            pairs()
        return fb

    def run(self, i, n_iter):
        while i < n_iter:
            if not self._have or i % self.gap == 0:
                if self.refresh() and n_iter - i > 1 and self.gap > 1:
                    with (self.capture_guard() if self.capture_guard else contextlib.nullcontext()):
                        i += self.batch.capture(steps_per_graph=self.gap - 1)     # one eager iteration, then the graph between refreshes
                    continue                                                       # (a refresh may be due at the new i)
            nxt = (i // self.gap + 1) * self.gap
            chunk = max(1, min(nxt, n_iter) - i)
            self.batch.step(chunk)
            i += chunk
        return i

    def report(self):
        return {"assign_refreshes": self.refreshes, "lap_fallbacks": self.fallbacks}


class OperatorLoop:
    """The reference's loop body (run_robot.py:154-221) over the HIP operators with PyTorch autograd and
    ``torch.optim.Adam``: the path of ``--model kinematic`` (and of a base model outside the fused engine).
    ``iteration(i)`` runs iteration i -- forward, assignment (re-solved on the GPU every ``assign_gap`` iterations) or
    Chamfer loss, optional flow loss, backward, Adam -- and returns the loss dictionary of tensors (no host sync)."""

    def __init__(self, args, model, cano_pc, pc_list, pc_ref_list=None, flow_ref_list=None, tau_func=None):
        self.args, self.model, self.cano_pc, self.pc_list = args, model, cano_pc, pc_list
        self.pc_ref_list, self.flow_ref_list, self.tau_func = pc_ref_list, flow_ref_list, tau_func
        device = cano_pc.device
        if args.model == "base":
            seg_params = [p for p in model.seg_head.parameters() if p.requires_grad]
            self.optimizer = torch.optim.Adam([{"params": [model.proposal_6d, model.proposal_t], "lr": args.trans_lr},
                                               {"params": seg_params, "lr": args.seg_lr}], lr=1e-3,
                                              weight_decay=args.weight_decay)
        else:
            self.optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=args.trans_lr,
                                              weight_decay=args.weight_decay)
        self.chamfer_dist = ChamferDistance()
        self.knn_flow = KNN(k=3, transpose_mode=True)
        self.matched = None
        # the kinematic model moves the cost matrices smoothly: the previous solve's pairs and potentials start the next
        # (shortest augmenting paths, reart_lap_resolve); the base model's resampled labels make them jump: cold solves
        self.lap_state = {} if args.model == "kinematic" else None
        self.lap_solves = self.lap_fallbacks = 0
        self.lap_events = None      # set to [] to collect a (start, end) torch.cuda.Event pair around every re-solve
        if args.use_assign_loss:
            # run_robot.py:167-169: both FPS calls sample fixed clouds (start 0 on the reference's CUDA path): once
            num_fps = pc_list.shape[1] // args.downsample
            zero = torch.zeros(1, dtype=torch.long, device=device)
            self.src_idx = farthest_point_sample(cano_pc[None], num_fps, start=zero, cuda_mode=True).expand(pc_list.shape[0], num_fps)
            self.tgt_pts = index_points(pc_list, farthest_point_sample(pc_list, num_fps, start=zero.expand(pc_list.shape[0]),
                                                                       cuda_mode=True))

    def iteration(self, i):
        from reart_amd.utils.lap import cdist, linear_sum_assignment_batch, linear_sum_assignment_points

        args, model, cano_pc, pc_list = self.args, self.model, self.cano_pc, self.pc_list
        kwargs = {"tau": self.tau_func(cur_iter=i + 1)} if args.model == "base" else {}
        pc_trans_list, seg_part, trans_list = model(cano_pc, **kwargs)
        losses, loss = {}, 0
        if args.use_assign_loss and i >= args.assign_iter:
            pc_src = index_points(pc_trans_list, self.src_idx)
            if self.matched is None or i % args.assign_gap == 0:  # run_robot.py:165-178
                # certified optimum on the GPU; an uncertified matrix falls back to scipy on the host, so this is
                # always the assignment the reference's linear_sum_assignment / parallel_lap returns
                if self.lap_events is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                if self.lap_state is not None:      # slowly moving problems: re-solve from the previous optimum
                    cols, fb, self.lap_stats = linear_sum_assignment_points(pc_src.detach(), self.tgt_pts, self.lap_state,
                                                                            return_stats="full", device_cols=True)
                else:
                    from reart_amd.utils import lap as lap_

                    st_ = {} if lap_.CANONICAL_TIES else None      # --deterministic: the potentials the tie check needs
                    assign, fb = linear_sum_assignment_batch(cdist(pc_src.detach(), self.tgt_pts), points=(pc_src.detach(), self.tgt_pts),
                                                             race=True, return_stats=True, state=st_, warm_assignment=st_ is not None)
                    if st_ is not None and lap_.canonicalize(pc_src.detach().contiguous(), self.tgt_pts.contiguous(), st_):
                        rows_ = np.arange(pc_src.shape[1], dtype=np.int64)
                        assign = [(rows_, c_) for c_ in st_["cols"].cpu().numpy().astype(np.int64)]
                if self.lap_events is not None:
                    ev[1].record()
                    self.lap_events.append(ev)
                self.lap_solves += 1
                self.lap_fallbacks += fb
                if self.lap_state is None:
                    cols = torch.from_numpy(np.stack([c for _, c in assign])).to(cano_pc.device)    # rows are 0..n-1 in order
                self.matched = self.tgt_pts.gather(1, cols[..., None].expand(-1, -1, 3))
            ass = args.lambda_assign * ((pc_src - self.matched) ** 2).sum(-1).sum()
            losses["opt assignment loss"] = ass
            loss = loss + ass
        else:
            rec = recon_loss(pc_trans_list, pc_list, self.chamfer_dist)
            losses["recon Loss"] = rec
            loss = loss + rec
        if args.use_flow_loss:
            c = args.cano_idx
            with torch.no_grad():
                comp = torch.cat((pc_trans_list[:c], cano_pc[None], pc_trans_list[c:]), dim=0)
                blended = [blend_anchor_motion(q, r, f, self.knn_flow, return_mask=True)
                           for q, r, f in zip(comp[:-1], self.pc_ref_list, self.flow_ref_list)]
            comp = torch.cat((pc_trans_list[:c], cano_pc[None], pc_trans_list[c:]), dim=0)
            fl = args.lambda_flow * flow_loss(torch.stack([b[0] for b in blended]), comp[1:] - comp[:-1],
                                              flow_mask_list=torch.stack([b[1] for b in blended]),
                                              robust=args.use_robust_loss)
            losses["flow Loss"] = fl
            loss = loss + fl
        losses["total Loss"] = loss
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return losses


def make_projection_loop(args, model, cano_pc, pc_list, pc_ref_list=None, flow_ref_list=None, tau_func=None):
    """The loop object for phase 2: the autograd-free ``KinematicEngine`` for every ``--model kinematic`` run -- the projection
    as the reference's README runs it (``--use_assign_loss --assign_iter 0``), its Chamfer branch (iterations before
    ``assign_iter``, runs without ``--use_assign_loss``), mixed joint types and root motion (the SAPIEN / real-scan variant of
    the model, networks/model.py:113-166); the base model outside the fused engine goes through ``OperatorLoop`` (PyTorch
    autograd + torch.optim.Adam over the same operators), which also remains the engine's reference in the tests."""
    if args.model == "kinematic":
        from reart_amd.kinematic_engine import KinematicEngine

        return KinematicEngine(model, cano_pc, pc_list, args.cano_idx, pc_ref_list if args.use_flow_loss else None,
                               flow_ref_list if args.use_flow_loss else None, trans_lr=args.trans_lr,
                               weight_decay=args.weight_decay, assign_iter=args.assign_iter, assign_gap=args.assign_gap,
                               downsample=args.downsample, lambda_assign=args.lambda_assign, lambda_flow=args.lambda_flow,
                               use_robust_loss=args.use_robust_loss, use_assign_loss=args.use_assign_loss)
    return OperatorLoop(args, model, cano_pc, pc_list, pc_ref_list, flow_ref_list, tau_func)


class SnapshotPrinter:
    """What the reference prints at ``i % snapshot_gap == 0`` and at the last iteration (run_robot.py:224-266): the loss line
    (the reference prints it every iteration, :216; here with the snapshot -- the fused loop does not meet the host in between)
    and, when the sample carries ground truth, `Flow eval: EPE | Acc 5 | Acc 10 | Angle`, `Seg eval: RI`, `Recon eval: recon`
    with the reference's formats and centimetre scaling.  The metrics are taken on a forward of the model as it stands after
    iteration i (the reference uses the forward iteration i itself ran: one Adam step and one noise draw earlier).
    ``lines`` keeps the metrics of every snapshot; ``out``: where the lines go (default stdout)."""

    GRAPH = os.environ.get("REART_SNAPSHOT_GRAPH", "1") != "0"

    def __init__(self, args, model, cano_pc, pc_list, sample=None, tau_func=None, out=None, graph=None):
        self.args, self.model, self.cano_pc, self.pc_list, self.sample = args, model, cano_pc, pc_list, sample
        self.tau_func, self.out = tau_func, out
        self.count, self.lines = 0, []
        # the metrics are ~80 small launches (elementwise / reductions / one matrix product): from the third snapshot on they are
        # ONE replay of a captured graph on copies of (seg_part, trans_list) -- README.md:125 prints 1 501 snapshots per run
        self.graph = self.GRAPH if graph is None else bool(graph)
        self._g = self._g_in = self._g_out = self._g_names = None
        # the ground truth the metrics compare with, on the device once (tail.snapshot_metrics would upload it at every snapshot)
        if sample is not None:
            keys = ("gt_flow_list", "gt_cano_part", "complete_gt_pc_list")
            self.sample = {k: torch.as_tensor(sample[k]).to(cano_pc.device) for k in keys if k in sample}

    def state(self, i):
        with torch.no_grad():
            if self.args.model == "base":
                _, seg_part, trans_list = self.model(self.cano_pc, tau=self.tau_func(cur_iter=i + 1))
            else:
                _, seg_part, trans_list = self.model(self.cano_pc)
        return seg_part, trans_list.detach()

    def __call__(self, i, losses):
        from reart_amd import tail

        import sys
        out = self.out if self.out is not None else sys.stdout
        if losses:
            print(f"iteration: {i} | " + " | ".join(f"{k}: {float(v.detach() if torch.is_tensor(v) else v):.3f}" for k, v in losses.items()),
                  file=out)
        seg_part, trans_list = self.state(i)
        has_gt = self.sample is not None and any(k in self.sample for k in ("gt_flow_list", "gt_cano_part", "complete_gt_pc_list"))
        m = self._metrics(seg_part, trans_list) if has_gt else {}
        if "epe" in m:
            print(f"Flow eval: EPE: {m['epe']:.3f} | Acc 5: {m['acc5']:.3f} | Acc 10: {m['acc10']:.3f} | Angle: {m['angle']:.3f}", file=out)
        if "ri" in m:
            print(f"Seg eval: RI: {m['ri']:.3f}", file=out)
        if "recon_err" in m:
            print(f"Recon eval: recon: {m['recon_err']:.3f}", file=out)
        self.count += 1
        self.lines.append((i, m))
        return m

    def _metrics(self, seg_part, trans_list):
        from reart_amd import tail

        c = self.args.cano_idx
        if not self.graph or self.count < 2:
            return tail.snapshot_metrics(self.cano_pc, self.pc_list, seg_part, trans_list, c, self.sample, chamfer=False)
        if self._g is None:
            try:
                self._g_in = (seg_part.clone(), trans_list.clone())
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    self._g_names, self._g_out = tail.snapshot_values(self.cano_pc, self._g_in[0], self._g_in[1], c, self.sample)
                self._g = g
            except Exception:           # a capture the runtime refuses: the eager form from here on
                self.graph, self._g = False, None
                torch.cuda.synchronize()
                return tail.snapshot_metrics(self.cano_pc, self.pc_list, seg_part, trans_list, c, self.sample, chamfer=False)
        if seg_part.shape != self._g_in[0].shape or trans_list.shape != self._g_in[1].shape:
            return tail.snapshot_metrics(self.cano_pc, self.pc_list, seg_part, trans_list, c, self.sample, chamfer=False)
        self._g_in[0].copy_(seg_part)
        self._g_in[1].copy_(trans_list)
        self._g.replay()
        return tail.snapshot_scaled(self._g_names, self._g_out.cpu().tolist()) if self._g_out is not None else {}


def build_kinematic_from_base(result, cano_pc, pc_list, args):
    """run_robot.py:101-124: KinematicModel from a base result (dict with pred_cano_part, pred_pose_list and, when the
    base run already extracted it, joint_connection)."""
    from reart_amd import tail

    device = cano_pc.device
    seg_part = torch.from_numpy(np.asarray(result["pred_cano_part"])).long().to(device)
    trans_list = torch.from_numpy(np.asarray(result["pred_pose_list"])).float().to(device)
    if "joint_connection" in result:
        joint_connection = torch.from_numpy(np.array(result["joint_connection"])).long().to(device)
    else:
        seg_part, trans_list, joint_connection = tail.extract_structure(
            seg_part, trans_list, cano_pc, merge_thr=args.merge_thr, cano_dist_thr=args.cano_dist_thr,
            lambda_joint=args.lambda_joint, min_num=0)
    new_seg, kin_kwargs = tail.kinematic_init(seg_part, trans_list, joint_connection)
    return KinematicModel(pose_len=pc_list.shape[0], seg_part=new_seg, cano_pc=cano_pc,
                          knn=KNN(k=1, transpose_mode=True), **kin_kwargs)


def main(args):
    torch.cuda.manual_seed_all(args.manual_seed)
    torch.manual_seed(args.manual_seed)
    np.random.seed(args.manual_seed)
    random.seed(args.manual_seed)
    if not torch.cuda.is_available():
        raise SystemExit("reart_amd runs on an AMD GPU only (no CPU fallback)")
    from reart_amd.utils import lap as _lap

    _lap.CANONICAL_TIES = bool(getattr(args, "deterministic", True))
    if args.evaluate and args.resume is None:
        raise ValueError("need model path to evaluate!")      # run_robot.py:86-87
    device = torch.device("cuda")
    dataset = None
    if args.synthetic:
        sample = synthetic_sequence(args.num_points, args.cano_idx, args.synthetic_frames, args.use_flow_loss)
    else:
        from reart_amd.dataset import Sequence

        dataset = Sequence(args.seq_path, num_points=args.num_points, cano_idx=args.cano_idx)
        sample = dataset[0]
    cano_pc = torch.from_numpy(sample["cano_pc"]).float().to(device)
    pc_list = torch.from_numpy(sample["pc_list"]).float().to(device)
    # --synthetic never shares a directory with a real sequence (seq_path keeps its default then)
    save_dir = os.path.join(args.save_root, "synthetic" if args.synthetic
                            else (os.path.basename(args.seq_path.rstrip("/")) or "sequence"))
    os.makedirs(save_dir, exist_ok=True)

    pc_ref_list = flow_ref_list = None
    if args.use_flow_loss:
        pc_ref_list, flow_ref_list = flow_references(args, sample, device, args.seq_path)

    tau_func = functools.partial(tau_cosine, max_iter=args.n_iter, end_temp=args.end_tau, start_temp=args.start_tau)
    fixed_tau = 0.0
    if args.model == "base":
        model = BaseModel(num_parts=args.num_parts, pose_len=pc_list.shape[0])
        if args.resume is not None:
            ckpt = torch.load(args.resume[0], map_location=device, weights_only=False)
            model.load_state_dict(ckpt["state_dict"], strict=False)
            fixed_tau = float(ckpt["tau"])  # the reference freezes tau on resume (:96-97)
            tau_func = lambda cur_iter: fixed_tau
            assert args.cano_idx == ckpt.get("cano_idx", args.cano_idx)
    else:
        if args.resume is None:  # run_robot.py:101-124: joint tree from a base result
            assert args.base_result_path is not None, "--model kinematic needs --base_result_path or --resume"
            with open(args.base_result_path, "rb") as f:
                result = pickle.load(f)
            print(f"load base result from {args.base_result_path}")
            assert args.cano_idx == result["cano_idx"]
            model = build_kinematic_from_base(result, cano_pc, pc_list, args)
        else:
            ckpt = torch.load(args.resume[0], map_location=device, weights_only=False)
            model = KinematicModel(pose_len=pc_list.shape[0], seg_part=ckpt["seg_part"].to(device),
                                   cano_pc=ckpt["cano_pc"].to(device), knn=KNN(k=1, transpose_mode=True),
                                   edge_index=ckpt["edge_index"], paths_to_base=ckpt["paths_to_base"],
                                   reverse_topo=ckpt["reverse_topo"])
            model.load_state_dict(ckpt["state_dict"], strict=True)
    model.to(device)
    chamfer_dist = ChamferDistance()
    knn_flow = KNN(k=3, transpose_mode=True)

    snapshot = SnapshotPrinter(args, model, cano_pc, pc_list, None if args.synthetic and "gt_flow_list" not in sample else sample, tau_func)

    n_iter = 1 if args.evaluate else args.n_iter
    i = 0
    lap_report = None       # assignment problems solved in the loop / how many of them went to the host solver
    # ---- phase 1: fused engine (base model, Chamfer [+ flow]) until the assignment loss takes over
    fused_until = n_iter if not args.use_assign_loss else min(args.assign_iter, n_iter)
    if args.model == "base" and not args.evaluate:
        eng = RelaxEngine(cano_pc, pc_list, model, args.cano_idx, pc_ref_list, flow_ref_list, n_iter=args.n_iter,
                          start_tau=args.start_tau, end_tau=args.end_tau, trans_lr=args.trans_lr, seg_lr=args.seg_lr,
                          lambda_flow=args.lambda_flow, use_robust_loss=args.use_robust_loss, fixed_tau=fixed_tau,
                          seed=args.manual_seed, weight_decay=args.weight_decay)
        if fused_until > 0:
            i += eng.capture()
        while i < fused_until:
            chunk = min(args.snapshot_gap, fused_until - i)
            eng.step(chunk)
            i += chunk
            row = eng.last_losses().cpu().numpy()
            snapshot(i - 1, {"recon Loss": row[0], "flow Loss": row[1], "total Loss": row[2]})
        # ---- phase 2 (base model): assignment loss on the same engine (run_robot.py:164-187)
        if i < n_iter and args.use_assign_loss:
            phase = AssignmentPhase(eng, cano_pc, pc_list, args.downsample, args.assign_gap, args.lambda_assign)

            def on_snapshot(k):
                row = eng.last_losses().cpu().numpy()
                snapshot(k - 1, {"opt assignment loss": row[0], "flow Loss": row[1], "total Loss": row[2]})

            i = phase.run(i, n_iter, args.snapshot_gap, on_snapshot)
            lap_report = phase.report()
            print(f"assignment phase: {lap_report['assign_refreshes']} refreshes, {lap_report['lap_fallbacks']} host fallbacks")
    # ---- phase 2 (kinematic model; base model without the engine): the reference's loop with HIP operators
    if i < n_iter and not args.evaluate:
        loop = make_projection_loop(args, model, cano_pc, pc_list, pc_ref_list, flow_ref_list, tau_func)
        while i < n_iter:
            losses = loop.iteration(i)
            if i % args.snapshot_gap == 0 or i == n_iter - 1:
                snapshot(i, losses)
            i += 1
        if args.use_assign_loss:
            lap_report = {"assign_refreshes": loop.lap_solves, "lap_fallbacks": loop.lap_fallbacks}
    if args.evaluate:
        snapshot(0, {})
    finish(args, model, cano_pc, pc_list, sample, save_dir, tau_func(cur_iter=n_iter), dataset, lap_report)
    print("all done!")
    return model


def finish(args, model, cano_pc, pc_list, sample, save_dir, tau, dataset=None, lap_report=None):
    """run_robot.py:227-356: structure, energies, result files (the reference's file names and keys)."""
    from reart_amd import tail
    from reart_amd.utils.kinematic_utils import edge_index2edges
    from reart_amd.utils.model_utils import compute_pc_transform

    device = cano_pc.device
    with torch.no_grad():
        _, seg_part, trans_list = model(cano_pc)
    trans_list = trans_list.detach()
    if isinstance(model, KinematicModel):   # the tree is the model's own (run_robot.py:235-237)
        from reart_amd.utils.graph_utils import denoise_seg_label

        seg_part = denoise_seg_label(seg_part.clone(), cano_pc, KNN(k=1, transpose_mode=True), min_num=20)
        conn = torch.tensor(edge_index2edges(model.edge_index), dtype=torch.long, device=device)
        from reart_amd.utils.kinematic_utils import extract_kinematic

        seg_part, trans_list, conn = extract_kinematic(seg_part, trans_list, conn)
    else:
        seg_part, trans_list, conn = tail.extract_structure(
            seg_part, trans_list, cano_pc, merge_thr=args.merge_thr, merge_it=args.merge_it,
            cano_dist_thr=args.cano_dist_thr, lambda_joint=args.lambda_joint)
    conn_list = conn.cpu().numpy().tolist()
    print("parts:", trans_list.shape[1], "| joint connection:", conn_list)
    metrics = tail.snapshot_metrics(cano_pc, pc_list, seg_part, trans_list, args.cano_idx, sample)
    if "epe" in metrics:
        print(f"Flow eval: EPE: {metrics['epe']:.3f} | Acc 5: {metrics['acc5']:.3f} | Acc 10: {metrics['acc10']:.3f} | "
              f"Angle: {metrics['angle']:.3f}")
    if "ri" in metrics:
        print(f"Seg eval: RI: {metrics['ri']:.3f}")
    if "recon_err" in metrics:
        print(f"Recon eval: recon: {metrics['recon_err']:.3f}")
    retarget_err = 9999      # run_robot.py:287-291: retargeting to the sequence's novel poses (kinematic model only)
    if isinstance(model, KinematicModel) and dataset is not None and len(dataset.novel_pose_list):
        from reart_amd.utils.kinematic_utils import ik

        retarget_err = ik(dataset, model, device, verbose=False, vis=False)
    print("Retarget error: {:.3f}".format(retarget_err))
    with open(os.path.join(save_dir, "result.txt"), "w") as f_result:
        f_result.write(f"retarget_err: {retarget_err:.3f}\n")
        for k in ("recon_err", "epe", "acc5", "acc10", "angle", "ri", "cd_err"):
            if k in metrics:
                f_result.write(f"{k}: {metrics[k]:.3f}\n")
        if not args.evaluate:
            energy = tail.energy_terms(cano_pc, pc_list, seg_part, trans_list, conn, args.cano_idx)
            print(f"Energy eval: total: {energy['total_err']:.3f}")
            for k in ("ass_err", "screw_err", "group_err", "total_err"):
                print(f"{k}: {energy[k]:.3f}")
                f_result.write(f"{k}: {energy[k]:.3f}\n")
            print(f"cd_err: {metrics['cd_err']:.3f}")
        if lap_report is not None:      # additions: how many assignment problems of the loop went to the host solver (0 = none)
            for k in ("assign_refreshes", "lap_fallbacks"):
                f_result.write(f"{k}: {lap_report[k]}\n")
    if args.evaluate:
        return
    save_dict = {"pred_cano_part": seg_part.cpu().numpy(), "pred_pose_list": trans_list.cpu().numpy(),
                 "cano_idx": args.cano_idx, "joint_connection": conn_list}
    save_dict.update(sample)
    with open(os.path.join(save_dir, "result.pkl"), "wb") as f:
        pickle.dump(save_dict, f)
    model_dict = {"state_dict": model.state_dict(), "tau": tau, "cano_idx": args.cano_idx}
    if isinstance(model, KinematicModel):
        model_dict.update(seg_part=model.seg_part, cano_pc=model.cano_pc, edge_index=model.edge_index,
                          paths_to_base=model.paths_to_base, reverse_topo=model.reverse_topo)
    torch.save(model_dict, os.path.join(save_dir, "model.pth.tar"))
    print("saved", os.path.join(save_dir, "result.pkl"), "and", os.path.join(save_dir, "model.pth.tar"))


def build_parser():
    p = argparse.ArgumentParser(description="Robot (reart_amd)")
    # same flags and defaults as the reference, run_robot.py:362-420
    p.add_argument("--manual_seed", default=2, type=int)
    p.add_argument("--resume", type=str, nargs="+", metavar="PATH")
    p.add_argument("--evaluate", dest="evaluate", action="store_true")
    p.add_argument("--snapshot_gap", default=100, type=int)
    p.add_argument("--use_cuda", default=1, type=int)
    p.add_argument("--cano_idx", default=0, type=int)
    p.add_argument("--num_points", default=4096, type=int)
    p.add_argument("--seq_path", default="data/robot/nao", type=str)
    p.add_argument("--normalize_file", default="data/category_normalize_scale.pkl", type=str)
    p.add_argument("--start_tau", default=5, type=float)
    p.add_argument("--end_tau", default=1, type=float)
    p.add_argument("--seg_lr", default=1e-3, type=float)
    p.add_argument("--trans_lr", default=1e-2, type=float)
    p.add_argument("--weight_decay", default=0, type=float)
    p.add_argument("--n_iter", default=15000, type=int)
    p.add_argument("--assign_iter", default=5000, type=int)
    p.add_argument("--num_parts", default=20, type=int)
    p.add_argument("--model", default="base", type=str, choices=["base", "kinematic"])
    p.add_argument("--base_result_path", default=None, type=str)
    p.add_argument("--corr_model_path", default="pretrained/corr_model.pth.tar")
    p.add_argument("--use_flow_loss", action="store_true")
    p.add_argument("--use_robust_loss", action="store_true")
    p.add_argument("--use_assign_loss", action="store_true")
    p.add_argument("--use_nproc", action="store_true")
    p.add_argument("--downsample", default=4, type=int)
    p.add_argument("--assign_gap", default=5, type=int)
    p.add_argument("--lambda_assign", default=3e-1, type=float)
    p.add_argument("--lambda_flow", default=1, type=float)
    p.add_argument("--lambda_joint", default=100, type=float)
    p.add_argument("--cano_dist_thr", default=1e-2, type=float)
    p.add_argument("--merge_thr", default=3e-2, type=float)
    p.add_argument("--merge_it", default=2, type=int)
    p.add_argument("--save_root", default="exp", type=str)
    # additions
    p.add_argument("--synthetic", action="store_true", help="generated articulated sequence instead of --seq_path")
    p.add_argument("--synthetic_frames", default=20, type=int)
    p.add_argument("--deterministic", dest="deterministic", action="store_true", default=True,
                   help="(default) the assignment refreshes return a function of the cost matrix alone, like the reference's scipy "
                        "call (run_robot.py:172-176): when several assignments are optimal the raced GPU solvers return whichever "
                        "finished first, so every solve is checked for ties (reart_lap_ties) and a tied problem takes the "
                        "lexicographically smallest optimum -- two runs under one --manual_seed are the same run (run_robot.py:37-49)")
    p.add_argument("--no_deterministic", dest="deterministic", action="store_false",
                   help="skip the tie check (about 2 %% of a projection iteration): tied optima are settled by whichever racer wins")
    return p


if __name__ == "__main__":
    a = build_parser().parse_args()
    os.makedirs(a.save_root, exist_ok=True)
    main(a)
