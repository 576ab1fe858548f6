"""Multi-GPU sweep: independent optimisation instances sharded across ranks.

One optimisation *instance* = (sequence, cano_idx, seed); instances share nothing (the reference's
``main(args)`` is self-contained, run_robot.py:35-358) and the reference selects the canonical frame
"by the lowest energy" (README.md:60) -- so the natural multi-GPU job is the sweep over
``cano_idx`` x sequences.  Ranks own whole instances (static round-robin, no data-path
collective); the only exchange is one ``all_gather`` of a fixed-size float record per instance
(RCCL over xGMI when the backend is "nccl"; latency-bound: 64 B per instance), after which every
rank can take the arg-min.
"""
import contextlib
import os
from concurrent.futures import ThreadPoolExecutor

import torch
import torch.distributed as dist

# floats per instance (a fixed 16-float record):
# [instance id, cano_idx, recon loss, flow loss, total loss, iterations, failed, parts,
#  total_err, ass_err, screw_err, group_err, cd_err, 0, 0, 0]   (energies: run_robot.py:306-321; NaN when not computed)
RECORD = 16
E_TOTAL = 8


def _record(inst, spec, losses=None, done=0, failed=0, energy=None):
    nan = float("nan")
    rec = [inst, spec.get("cano_idx", -1)] + (list(losses) if losses is not None else [nan] * 3) + [done, failed]
    if energy is None:
        rec += [nan] * 6
    else:
        rec += [energy.get("parts", nan), energy["total_err"], energy["ass_err"], energy["screw_err"], energy["group_err"],
                energy.get("cd_err", nan)]
    return torch.tensor(rec + [0.0] * (RECORD - len(rec)), dtype=torch.float32)


def best_instance(records):
    """Index of the instance the reference would keep: lowest energy (README.md:60); the final loss decides when no
    energies were computed."""
    key = records[:, E_TOTAL].clone()
    if torch.isnan(key).all():
        key = records[:, 4].clone()
    key[torch.isnan(key)] = float("inf")
    return int(torch.argmin(key).item())


def instance_energy(eng, spec, **thresholds):
    """Structure + model-selection energy of a finished engine (reart_amd.tail) -> dict for the record."""
    from . import tail

    # caller-order clouds: the tail depends on point order (per-part FPS starts, FPS / assignment tie rules), so the
    # energy that picks the winning cano_idx is the value run_robot.finish computes for the same instance
    cano, pcs = eng.caller_clouds()
    res = tail.finish_instance(eng.model, cano, pcs, int(spec.get("cano_idx", eng.cfg.cano_idx)), **thresholds)
    res["parts"] = int(res["trans_list"].shape[1])
    return res


class _CaptureGate:
    """Graph capture on one thread and GPU work (allocation, frees, synchronisation) on others do not mix: on this stack a
    tensor freed by a worker thread while the main thread had a capture open ended the process (segmentation fault), even
    in thread-local capture mode.  The tails of finished instances therefore run as READERS of this gate and every capture
    as its one WRITER: replays and eager launches overlap the tails as before, a capture waits for the tails in flight."""

    def __init__(self):
        import threading

        self._cv, self._readers, self._writer, self._waiting = threading.Condition(), 0, False, 0

    def tail(self):
        gate = self

        class _R:
            def __enter__(self):
                with gate._cv:
                    while gate._writer or gate._waiting:      # a waiting capture goes first: readers that keep coming (the
                        gate._cv.wait()                       # refreshes of concurrent groups) must not starve it
                    gate._readers += 1

            def __exit__(self, *exc):
                with gate._cv:
                    gate._readers -= 1
                    gate._cv.notify_all()
        return _R()

    def capture(self):
        gate = self

        class _W:
            def __enter__(self):
                with gate._cv:
                    gate._waiting += 1
                    while gate._writer or gate._readers:
                        gate._cv.wait()
                    gate._waiting -= 1
                    gate._writer = True

            def __exit__(self, *exc):
                with gate._cv:
                    gate._writer = False
                    gate._cv.notify_all()
        return _W()


def shard(n_instances, rank, world):
    """Static round-robin assignment: instance i runs on rank i % world."""
    return list(range(rank, n_instances, world))


def instance_cost(spec):
    """Relative cost of one instance for the longest-first deal: frames x points^2 (the all-pairs searches dominate an
    iteration; SURVEY.md 8e "longest-first greedy by T*N^2").  Specs without the fields cost 1."""
    if "cost" in spec:
        return float(spec["cost"])
    n = float(spec.get("points", 1) or 1)
    return float(spec.get("frames", 1) or 1) * n * n


def deal(instances, world, policy="round_robin"):
    """-> list over ranks of the instance ids each rank owns.  Every rank computes the same deal from the same list, so no
    communication is needed.  ``round_robin``: instance i on rank i % world (right for equal instances: a cano_idx sweep of
    one sequence).  ``lpt``: longest processing time first -- instances sorted by ``instance_cost`` (ties: lower id), each
    to the rank with the least load so far (ties: lower rank); for sequence sets with different frame counts / sizes it
    bounds the slowest rank by 4/3 of the optimum where round-robin can be off by the largest instance per round."""
    n = len(instances)
    if policy == "round_robin":
        return [shard(n, r, world) for r in range(world)]
    if policy != "lpt":
        raise ValueError("policy is 'round_robin' or 'lpt'")
    cost = [instance_cost(s) for s in instances]
    load, out = [0.0] * world, [[] for _ in range(world)]
    for i in sorted(range(n), key=lambda i: (-cost[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        out[r].append(i)
        load[r] += cost[i]
    return out


def owner_of(plan):
    """{instance id: rank} of a deal."""
    return {i: r for r, ids in enumerate(plan) for i in ids}


def gather_records(local, n_instances, device, plan=None):
    """local: {instance id: 1-D float tensor [RECORD]} of this rank -> [n_instances, RECORD] on every
    rank, ordered by instance id.  Uses all_gather on padded per-rank blocks (equal message size).  ``plan``: the deal
    (``deal``) the ranks ran under; round-robin when omitted."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if plan is None:
        plan = [shard(n_instances, r, world) for r in range(world)]
    per = max([len(ids) for ids in plan] + [1])
    block = torch.full((per, RECORD), float("nan"), dtype=torch.float32, device=device)
    for slot, inst in enumerate(plan[rank]):
        block[slot] = local[inst].to(device=device, dtype=torch.float32)
    if world > 1:
        blocks = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(blocks, block)
    else:
        blocks = [block]
    out = torch.empty((n_instances, RECORD), dtype=torch.float32, device=device)
    for r in range(world):
        for slot, inst in enumerate(plan[r]):
            out[inst] = blocks[r][slot]
    return out


def run_sweep(instances, run_instance, device, policy="round_robin"):
    """instances: list of dicts (at least ``cano_idx``); ``run_instance(spec) -> dict(recon, flow,
    total, iterations)`` optimises one instance on this rank's GPU.  ``policy``: how instances are dealt to ranks
    (``deal``).  Returns (records [n, RECORD], index of the lowest-energy instance)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    plan = deal(instances, world, policy)
    local = {}
    for inst in plan[rank]:
        spec = instances[inst]
        try:
            res = run_instance(spec)
            local[inst] = _record(inst, spec, (res["recon"], res["flow"], res["total"]), res["iterations"], 0,
                                  res if "total_err" in res else None)
        except Exception:  # a failed instance is reported (NaN energy), it does not kill the job
            local[inst] = _record(inst, spec, failed=1)
    records = gather_records(local, len(instances), device, plan)
    return records, best_instance(records)


def run_sweep_engines(instances, make_engine, n_iter, device, per_gpu=3, chunk=100, energy=False, mode="streams",
                      overlap_tails=True, on_finish=None, policy="round_robin", assign=None, groups_in_flight=4):
    """Sweep of fused-loop instances with ``per_gpu`` of them in flight per GPU.

    One instance of the relaxation loop is a chain of short, latency-bound launches that leaves
    issue slots idle; independent instances on separate streams fill them (measured on MI355X:
    8.0 k it/s for one instance, 14.2 k it/s aggregate for three).  ``make_engine(spec)`` returns a
    prepared ``reart_amd.relax.RelaxEngine`` (its tensors live on ``device``); this rank's instances
    are optimised ``per_gpu`` at a time, stepped round-robin in graph replays of ``chunk``
    iterations.  ``energy=True`` finishes every instance with the reference's structure extraction and energy terms
    (``instance_energy``), which then decide the winner.  ``mode="batch"``: the instances of a group (same shape; up to
    ``RelaxBatch.MAX``) advance in SHARED launches (``reart_relax_step_batch``) instead of on one stream each -- the same
    results, and an aggregate rate that does not depend on how the runtime maps streams to hardware queues.
    A group whose engines the shared launches cannot take (different shapes or switches, a loss branch the batched entry
    does not implement) falls back to the streams path for that group; it never aborts the sweep.
    ``assign = dict(assign_iter, assign_gap, downsample, lambda_assign)``: the README recipe's second phase
    (run_robot.py:164-187, ``--use_assign_loss``) -- after ``assign_iter`` iterations the Chamfer loss gives way to the
    assignment loss, its pairs refreshed every ``assign_gap`` iterations; a batch group solves the problems of all its
    instances in one call per refresh (``run_robot.AssignmentPhaseBatch``) and keeps stepping in shared launches.
    ``on_finish(inst, spec, engine, energy dict or None)`` is called once per finished instance (the command line writes
    the instance's result files there).  ``groups_in_flight``: how many groups of the recipe sweep run concurrently (each
    with its engines, graphs, a stream and a host thread); the rest are built when a slot frees up.
    Returns (records [n, RECORD], best index) like ``run_sweep``."""
    if mode not in ("streams", "batch"):
        raise ValueError("mode is 'streams' or 'batch'")
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    plan_ranks = deal(instances, world, policy)
    mine = plan_ranks[rank]
    local = {}
    threads = os.environ.get("REART_SWEEP_THREADS", "1") != "0"
    pool, pending = None, []
    gate = _CaptureGate()
    # groups of (nearly) equal size, none larger than per_gpu: 20 instances at 6 per GPU run as 5 + 5 + 5 + 5, not 6 + 6 + 6 + 2
    # (a short last group leaves the chip to two instances: 18 k instead of 22 k it/s for its share)
    n_groups = (len(mine) + per_gpu - 1) // per_gpu if mine else 0
    size = (len(mine) + n_groups - 1) // n_groups if n_groups else 0
    groups = [mine[g0:g0 + size] for g0 in range(0, len(mine), size)] if size else []

    # A group goes through four stages, and the loop below runs them so that the GPU always has a group's iterations queued:
    #   build    host work (loading / generating the sequence, k-d orders) + the engines' set-up launches on their own streams
    #   capture  graphs of `chunk` iterations (shared launches: one per part of <= RelaxBatch.MAX engines); synchronises the device
    #   enqueue  every iteration of the group, asynchronously (graph replays); an event marks the end
    #   tails    structure + energy + result files per instance, on host threads
    # build(g+1) runs on the host while the GPU steps group g; capture(g+1) comes BEFORE the tails of group g are started
    # (captures and tails never overlap, _CaptureGate), so those tails then run under the iterations of group g+1.
    def build(group):
        live = []
        for inst in group:
            spec = instances[inst]
            st = torch.cuda.Stream(device=device)
            try:
                with torch.cuda.stream(st):
                    eng = make_engine(spec)
                live.append([inst, spec, eng, st, 0, None])           # [.., iterations queued, event after the last one]
            except Exception:  # a failed instance is reported (NaN energy), it does not kill the job
                local[inst] = _record(inst, spec, failed=1)
        return live

    def capture(live):
        """-> plan: list of (RelaxBatch, its entries, iterations its capture already ran) for the shared launches; entries not
        covered by a batch carry their own graph (streams)."""
        plan = []
        solo = list(live)
        if mode == "batch" and live:
            from .relax import RelaxBatch

            for e in live:
                e[3].synchronize()                       # the engines were prepared on their own streams
            solo = []
            dead = []                                     # entries whose shared launches failed after advancing somebody
            # the parts are cut ONCE from a snapshot: `live` itself is only edited after the loop
            for part in [live[b0:b0 + RelaxBatch.MAX] for b0 in range(0, len(live), RelaxBatch.MAX)]:
                try:
                    batch = RelaxBatch([e[2] for e in part])     # refuses engines that do not share shape and switches
                    # capture() runs its first step eagerly: an engine the batched entry does not implement
                    # (REART_ERR_UNSUPPORTED) shows up there, before a capture is open, and the part takes the streams path
                    used = 0
                    if first_phase > 1:
                        with gate.capture():
                            used = batch.capture(steps_per_graph=min(chunk, first_phase - 1))
                    plan.append((batch, part, used))
                except Exception as exc:
                    started = {int(e[2].iter.item()) for e in part}
                    if started != {0}:          # the shared launches already advanced somebody: not restartable here
                        for e in part:
                            local[e[0]] = _record(e[0], e[1], failed=1)
                            dead.append(e)
                    else:
                        import warnings

                        warnings.warn(f"sweep: batch of {len(part)} instances falls back to streams ({type(exc).__name__}: {exc})")
                        solo.extend(part)
            for e in dead:
                live.remove(e)
        for e in solo:
            try:
                with torch.cuda.stream(e[3]), gate.capture():
                    e[4] = e[2].capture(steps_per_graph=min(chunk, first_phase))
            except Exception as exc:  # a failed capture is this instance's failure (NaN energy), never the sweep's
                import sys

                print(f"sweep: instance {e[0]} ({e[1]}) failed in graph capture: {type(exc).__name__}: {exc}", file=sys.stderr)
                local[e[0]] = _record(e[0], e[1], failed=1)
                live.remove(e)
        return plan

    if assign is not None and int(assign["assign_iter"]) < 1:
        raise ValueError("the sweep's assignment phase starts after at least one Chamfer iteration (assign_iter >= 1); "
                         "run_robot.py handles assign_iter 0")
    first_phase = n_iter if assign is None else min(int(assign["assign_iter"]), n_iter)
    lap_counts = {"assign_refreshes": 0, "lap_fallbacks": 0}

    def enqueue(live, plan):
        batched = set()
        for batch, part, used in plan:
            batch.step(first_phase - used)
            if first_phase < n_iter:          # the assignment phase: host-driven (a refresh reads its certificates), shared launches
                from .run_robot import AssignmentPhaseBatch

                with (gate.tail() if concurrent else contextlib.nullcontext()):      # (allocates: not while another group captures)
                    ph = AssignmentPhaseBatch(batch, [e[2].caller_clouds() for e in part], assign["downsample"], assign["assign_gap"],
                                              assign["lambda_assign"])
                ph.capture_guard = gate.capture
                ph.work_guard = gate.tail if concurrent else None
                ph.run(first_phase, n_iter)
                lap_counts["assign_refreshes"] += ph.refreshes
                lap_counts["lap_fallbacks"] += ph.fallbacks
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            for e in part:
                e[4], e[5] = n_iter, ev
                batched.add(id(e))
        rest = [e for e in live if id(e) not in batched]
        while any(e[4] < first_phase for e in rest):     # streams: round-robin graph replays
            for e in rest:
                if e[4] < first_phase:
                    n = min(chunk, first_phase - e[4])
                    with torch.cuda.stream(e[3]):
                        e[2].step(n)
                    e[4] += n
        if first_phase < n_iter:                          # streams: the assignment phase, one instance after the other
            from .run_robot import AssignmentPhase

            for e in rest:
                with torch.cuda.stream(e[3]):
                    ph = AssignmentPhase(e[2], *e[2].caller_clouds(), assign["downsample"], assign["assign_gap"], assign["lambda_assign"])
                    ph.capture_guard = gate.capture
                    e[4] = ph.run(first_phase, n_iter)
                lap_counts["assign_refreshes"] += ph.refreshes
                lap_counts["lap_fallbacks"] += ph.fallbacks
        for e in rest:
            e[5] = torch.cuda.Event()
            e[5].record(e[3])

    def finish(entry):
        with gate.tail():
            return _finish(entry)

    def _finish(entry):
        inst, spec, eng, st, done, ev = entry
        ev.synchronize()                                  # the instance's last iteration (its own stream or the batch's)
        st.wait_event(ev)
        with torch.cuda.stream(st):
            row = eng.last_losses().cpu()
        en = None
        if energy:
            try:
                with torch.cuda.stream(st):
                    en = instance_energy(eng, spec)
            except Exception as exc:      # e.g. every part merged away: the losses still describe the instance
                import sys
                import traceback

                en = None
                print(f"sweep: instance {inst} ({spec}) has no energy: {type(exc).__name__}: {exc}\n"
                      + "".join(traceback.format_exception(type(exc), exc, exc.__traceback__)[-3:]), file=sys.stderr)
        if on_finish is not None:
            try:
                on_finish(inst, spec, eng, en)
            except Exception as exc:      # disk full, pickling error ...: the record still counts, the job goes on
                import sys

                print(f"sweep: on_finish of instance {inst} ({spec}) raised {type(exc).__name__}: {exc}", file=sys.stderr)
        return inst, _record(inst, spec, (float(row[0]), float(row[1]), float(row[2])), done, 0, en)

    def tails(live):
        # The end of an instance is latency-bound (its assignment solves occupy T-1 of the 256 compute units) and full of
        # host round trips: the instances of a group finish side by side, each on its own stream and host thread -- and
        # (overlap_tails) under the next group's iterations.
        nonlocal pool
        if energy and threads:
            if pool is None:
                pool = ThreadPoolExecutor(max_workers=2 * max(per_gpu, 1))
            futures = [pool.submit(finish, e) for e in live]
            if overlap_tails:
                pending.extend(futures)
                return futures
            for f in futures:
                inst, rec = f.result()
                local[inst] = rec
        else:
            for inst, rec in (finish(e) for e in live):
                local[inst] = rec
        return None

    import time

    stages = {"build_s": 0.0, "capture_s": 0.0, "drain_s": 0.0, "groups": len(groups)}   # host seconds per stage (this rank)

    def timed(key, fn, *a):
        t_ = time.perf_counter()
        r_ = fn(*a)
        stages[key] += time.perf_counter() - t_
        return r_

    # The README recipe as a sweep (assign): a group's second phase is driven from the host refresh by refresh, and while its
    # K x (T-1) assignment problems are being solved (one workgroup each, as long as the slowest of them) the rest of the chip
    # idles -- 65 s for 20 instances when the groups took turns.  The groups are independent: here every group runs BOTH
    # phases on a stream and a host thread of its own, all groups at once; the solves of one group overlap the solves and the
    # iterations of the others (the solver's racers and reduction workgroups are dealt with the number of concurrent calls
    # in mind, lap.CONCURRENT_CALLS).  Captures stay exclusive (the gate's one writer); refreshes and tails are its readers.
    concurrent = (assign is not None and first_phase < n_iter and mode == "batch" and threads and len(groups) > 1
                  and os.environ.get("REART_SWEEP_CONCURRENT", "1") != "0")
    if concurrent:
        import threading

        from .utils import lap as _lap

        # At most `groups_in_flight` groups hold engines, graphs and a host thread at a time (`per_gpu` bounds
        # the instances of a GROUP; with every group of a rank resident at once nothing bounded device memory, the thread
        # count or the solver's share of the chip).  The first wave is built and captured up front on this thread, as
        # measured (starting a group while the next is still being built: 25.9 against 24.7 s); every later group is built
        # and captured lazily by the worker that takes it, inside its turn -- the build as a READER of the gate (it
        # allocates), the capture as its writer -- after that worker's previous group has finished its tails and let go of
        # its engines.
        bound = max(1, min(int(groups_in_flight), len(groups)))
        first = [timed("build_s", build, g) for g in groups[:bound]]
        ready = [(live, timed("capture_s", capture, live)) for live in first]
        del first
        later = list(range(bound, len(groups)))
        if energy and pool is None:
            pool = ThreadPoolExecutor(max_workers=2 * max(per_gpu, 1))
        errors = []
        take = threading.Lock()
        peak = {"now": 0, "max": 0}

        def drive(k):
            live, plan = ready[k]
            ready[k] = None
            gi = k
            while True:
                try:
                    with take:
                        peak["now"] += 1
                        peak["max"] = max(peak["max"], peak["now"])
                    st = torch.cuda.Stream(device=device)
                    st.wait_stream(torch.cuda.default_stream(device))      # the set-up and capture work of the main thread
                    with torch.cuda.stream(st):
                        enqueue(live, plan)
                    for f in tails(live) or ():           # this group's tails end before the worker takes another group
                        f.exception()
                except BaseException as exc:      # the group's instances are reported failed; the other groups go on
                    errors.append((gi, exc))
                    for e in live:
                        local.setdefault(e[0], _record(e[0], e[1], failed=1))
                finally:
                    with take:
                        peak["now"] -= 1
                live = plan = None                        # engines and graphs of the finished group go back to the allocator
                with take:
                    if not later:
                        return
                    gi = later.pop(0)
                try:
                    with gate.tail():
                        live = build(groups[gi])
                    plan = capture(live)
                except BaseException as exc:
                    errors.append((gi, exc))
                    for inst in groups[gi]:
                        local.setdefault(inst, _record(inst, instances[inst], failed=1))
                    live, plan = [], []

        old_calls = _lap.CONCURRENT_CALLS
        _lap.CONCURRENT_CALLS = bound
        try:
            workers = [threading.Thread(target=drive, args=(k,), name=f"sweep-group-{k}") for k in range(bound)]
            for w in workers:
                w.start()
            for w in workers:
                w.join()
        finally:
            _lap.CONCURRENT_CALLS = old_calls
        for k, exc in errors:
            import sys

            print(f"sweep: group {k} failed: {type(exc).__name__}: {exc}", file=sys.stderr)
        stages["concurrent_groups"] = bound
        stages["groups_in_flight_max"] = peak["max"]
        groups = []                                        # nothing left for the pipelined loop below
    cur = timed("build_s", build, groups[0]) if groups else []
    enqueue(cur, timed("capture_s", capture, cur))
    for gi in range(len(groups)):
        nxt = timed("build_s", build, groups[gi + 1]) if gi + 1 < len(groups) else None      # host work under the GPU's stepping of `cur`
        nplan = timed("capture_s", capture, nxt) if nxt is not None else None   # waits for the tails in flight and for `cur`'s iterations
        if nxt is not None and not (energy and threads):
            enqueue(nxt, nplan)                                             # inline tails: keep the GPU busy while the host reads results
            tails(cur)
        else:
            tails(cur)
            if nxt is not None:
                enqueue(nxt, nplan)
        cur = nxt
    t_ = time.perf_counter()
    for f in pending:
        inst, rec = f.result()
        local[inst] = rec
    if pool is not None:
        pool.shutdown()
    torch.cuda.synchronize(device)
    stages["drain_s"] = time.perf_counter() - t_          # what is left after the last group was queued: its iterations and tails
    stages.update(lap_counts)
    run_sweep_engines.last_stages = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in stages.items()}
    records = gather_records(local, len(instances), device, plan_ranks)
    return records, best_instance(records)


# ------------------------------------------------------------------------------------------------------------------
# Command line: BASELINE.json configs[3] -- "all robot categories x cano_idx sweep sharded across 8 MI355X".
#
#   python -m reart_amd.sweep --seq_root data/robot --cano all --n_iter 15000 --energy --gpus 8 --save_root exp/sweep
#
# The reference has no sweep driver: it is run once per (sequence, cano_idx) by hand (`main(args)`, run_robot.py:35-358) and
# "the canonical frame index is selected by the lowest energy" (README.md:58-60) from the printed energies
# (run_robot.py:306-321).  This is that procedure as one job: enumerate, shard round-robin over the ranks, optimise, gather
# one 16-float record per instance, take the arg-min per sequence, keep the winner's result files.

def list_sequences(seq_root, names=None):
    """Sequence directories under ``seq_root`` in the reference's layout (``state_0.pkl`` + ``pose_i.pkl``,
    dataset/dataset_robot.py:14-22) -> [(name, path, number of frames)], sorted by name."""
    import glob

    out = []
    for d in sorted(os.listdir(seq_root)):
        path = os.path.join(seq_root, d)
        if names and d not in names:
            continue
        if os.path.isdir(path) and os.path.exists(os.path.join(path, "state_0.pkl")):
            out.append((d, path, 1 + len(glob.glob(os.path.join(path, "pose_*.pkl")))))
    if names:
        missing = sorted(set(names) - {n for n, _, _ in out})
        if missing:
            raise FileNotFoundError(f"no sequence directory for {missing} under {seq_root}")
    return out


def enumerate_instances(sequences, cano="all"):
    """[(name, path, T)] x canonical indices -> instance specs in a fixed order (sequence-major): every rank enumerates
    the same list, so the round-robin shard needs no communication."""
    out = []
    for name, path, T in sequences:
        idxs = range(T) if cano == "all" else [int(c) for c in str(cano).split(",")]
        for c in idxs:
            if not 0 <= c < T:
                raise ValueError(f"cano_idx {c} outside the {T} frames of {name}")
            out.append({"seq": name, "seq_path": path, "cano_idx": int(c), "frames": T})
    return out


def winners(instances, records):
    """Per sequence: the instance with the lowest total energy (the final loss when no energies were computed) ->
    {sequence name: index into ``instances``}; a sequence whose instances all failed maps to None."""
    rec = torch.as_tensor(records).float().cpu()
    out = {}
    for name in dict.fromkeys(s["seq"] for s in instances):
        ids = [i for i, s in enumerate(instances) if s["seq"] == name]
        sub = rec[ids]
        ok = ~torch.isnan(sub[:, E_TOTAL] if not torch.isnan(sub[:, E_TOTAL]).all() else sub[:, 4])
        out[name] = ids[best_instance(sub)] if bool(ok.any()) else None
    return out


def build_cli():
    import argparse

    p = argparse.ArgumentParser(prog="python -m reart_amd.sweep", description="(sequence x cano_idx) sweep, one rank per GPU")
    p.add_argument("--seq_root", default="data/robot", help="directory holding one sub-directory per sequence / category")
    p.add_argument("--seqs", default="", help="comma-separated sub-directory names (default: all)")
    p.add_argument("--cano", default="all", help="'all' or comma-separated canonical frame indices")
    p.add_argument("--synthetic", type=int, default=0, metavar="K", help="K generated sequences instead of --seq_root")
    p.add_argument("--synthetic_frames", type=int, default=20)
    p.add_argument("--gpus", type=int, default=1, help="ranks to start when not already under torch.distributed.run")
    p.add_argument("--per_gpu", type=int, default=6, help="instances in flight per GPU")
    p.add_argument("--groups_in_flight", type=int, default=4,
                   help="recipe sweep (--use_assign_loss): groups of --per_gpu instances that run concurrently on one GPU, each on a "
                        "stream and host thread of its own; further groups are built when one finishes")
    p.add_argument("--mode", choices=("streams", "batch"), default="batch",
                   help="batch (default): the instances of a group advance in shared launches (24.5 k it/s per GPU for six at "
                        "T=20 x N=4096 against 12.3 k for one); streams: one stream per instance (a group the shared launches "
                        "cannot take -- mixed shapes -- falls back to it by itself)")
    p.add_argument("--energy", action="store_true", help="end every instance with structure extraction + energy (run_robot.py:224-321)")
    p.add_argument("--save_root", default="exp/sweep")
    p.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    p.add_argument("--shard", choices=("round_robin", "lpt"), default="round_robin",
                   help="how instances are dealt to ranks: round_robin (instance i on rank i %% world) or lpt = longest first by "
                        "frames x points^2 onto the least loaded rank (sequence sets of unequal length)")
    # the per-instance flags of run_robot.py:362-420, same names and defaults
    p.add_argument("--manual_seed", default=2, type=int)
    p.add_argument("--num_points", default=4096, type=int)
    p.add_argument("--num_parts", default=20, type=int)
    p.add_argument("--n_iter", default=15000, type=int)
    p.add_argument("--start_tau", default=5, type=float)
    p.add_argument("--end_tau", default=1, type=float)
    p.add_argument("--seg_lr", default=1e-3, type=float)
    p.add_argument("--trans_lr", default=1e-2, type=float)
    p.add_argument("--weight_decay", default=0, type=float)
    p.add_argument("--use_flow_loss", action="store_true")
    p.add_argument("--use_robust_loss", action="store_true")
    p.add_argument("--lambda_flow", default=1, type=float)
    p.add_argument("--use_assign_loss", action="store_true", help="README.md:116: assignment loss after --assign_iter iterations")
    p.add_argument("--assign_iter", default=5000, type=int)
    p.add_argument("--assign_gap", default=5, type=int)
    p.add_argument("--downsample", default=4, type=int)
    p.add_argument("--lambda_assign", default=3e-1, type=float)
    p.add_argument("--corr_model_path", default="pretrained/corr_model.pth.tar")
    p.add_argument("--normalize_file", default="data/category_normalize_scale.pkl", type=str)
    p.add_argument("--deterministic", dest="deterministic", action="store_true", default=True,
                   help="(default) assignment refreshes settle tied optima canonically (run_robot.py --deterministic): a sweep repeats run to run")
    p.add_argument("--no_deterministic", dest="deterministic", action="store_false")
    return p


def _engine_factory(args, device, samples):
    """make_engine(spec) for run_sweep_engines: one BaseModel + RelaxEngine per instance, exactly what
    ``run_robot.main`` builds for ``--model base`` (run_robot.py:91-98, 145-151)."""
    from .networks.model import BaseModel
    from .relax import RelaxEngine
    from . import run_robot as rr

    ref_cache = {}        # sequence -> (pc_ref_list, flow_ref_list): they depend on the sequence only, not on cano_idx

    def references(spec, sample):
        key = spec.get("seq_path") or ("synthetic", spec.get("synthetic"))
        if spec.get("synthetic") is not None:      # the generator's own references travel with the sample
            return rr.flow_references(args, sample, device, None)
        if key not in ref_cache:
            ref_cache[key] = rr.flow_references(args, sample, device, spec.get("seq_path"))
        return ref_cache[key]

    def make_engine(spec):
        if spec.get("synthetic") is not None:
            sample = rr.synthetic_sequence(args.num_points, spec["cano_idx"], spec["frames"], args.use_flow_loss,
                                           seed=spec["synthetic"])
        else:
            sample = rr.load_sequence(spec["seq_path"], args.num_points, spec["cano_idx"])
        cano = torch.from_numpy(sample["cano_pc"]).float().to(device)
        pcs = torch.from_numpy(sample["pc_list"]).float().to(device)
        refs = flows = None
        if args.use_flow_loss:
            refs, flows = references(spec, sample)
        torch.manual_seed(args.manual_seed)                   # run_robot.py:36-41: every run seeds the seg-head init alike
        model = BaseModel(num_parts=args.num_parts, pose_len=pcs.shape[0]).to(device)
        samples[spec["id"]] = sample
        return RelaxEngine(cano, pcs, model, spec["cano_idx"], refs, flows, n_iter=args.n_iter, start_tau=args.start_tau,
                           end_tau=args.end_tau, trans_lr=args.trans_lr, seg_lr=args.seg_lr, lambda_flow=args.lambda_flow,
                           use_robust_loss=args.use_robust_loss, seed=args.manual_seed, weight_decay=args.weight_decay)

    return make_engine


def _engine_poses(eng):
    """[T-1, P, 4, 4] poses of a finished engine's model (BaseModel.forward's third output, networks/model.py:39-70)."""
    with torch.no_grad():
        return eng.model(eng.caller_clouds()[0])[2].detach()


def instance_dir(save_root, spec):
    return os.path.join(save_root, spec["seq"], f"cano_{spec['cano_idx']}")


def save_instance(save_root, spec, result, sample=None, model=None, tau=None):
    """result.pkl (+ model.pth.tar) of one instance with the reference's keys (run_robot.py:333-356)."""
    import pickle

    import numpy as np

    d = instance_dir(save_root, spec)
    os.makedirs(d, exist_ok=True)
    to_np = lambda x: x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    out = {"pred_cano_part": to_np(result["seg_part"]), "pred_pose_list": to_np(result["trans_list"]),
           "cano_idx": int(spec["cano_idx"]), "joint_connection": to_np(result["joint_connection"]).tolist()}
    if sample is not None:
        out.update(sample)
    with open(os.path.join(d, "result.pkl"), "wb") as f:
        pickle.dump(out, f)
    if model is not None:
        torch.save({"state_dict": model.state_dict(), "tau": tau, "cano_idx": int(spec["cano_idx"])},
                   os.path.join(d, "model.pth.tar"))
    return d


def main(argv=None, runner=None):
    """The sweep job.  ``runner(spec) -> dict`` replaces the GPU engine per instance (tests run the CPU oracle through it
    under gloo); the product path (``runner is None``) needs a GPU per rank and raises without one."""
    import json
    import shutil
    import sys

    from . import launch

    import time

    t_start = time.perf_counter()
    args = build_cli().parse_args(argv)
    if args.gpus > 1 and not launch.under_launcher():
        # N ranks of this module, started before anything here has touched the GPU
        return launch.self_launch("reart_amd.sweep", list(sys.argv[1:] if argv is None else argv), args.gpus, module=True)
    world = launch.check_world(args.gpus) if launch.under_launcher() else 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if runner is None:
        if not torch.cuda.is_available():
            raise SystemExit("reart_amd.sweep needs an MI355X per rank: the HIP path has no CPU fallback")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        from .utils import lap as _lap

        _lap.CANONICAL_TIES = bool(args.deterministic)
    else:
        device = torch.device("cpu")
    if launch.under_launcher() and not dist.is_initialized():
        backend = args.backend or ("nccl" if device.type == "cuda" else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, **kw)

    if args.synthetic:
        sequences = [(f"synthetic_{k}", None, args.synthetic_frames) for k in range(args.synthetic)]
    else:
        names = [s for s in args.seqs.split(",") if s]
        sequences = list_sequences(args.seq_root, names or None)
    if not sequences:
        raise SystemExit(f"no sequences under {args.seq_root}")
    instances = enumerate_instances(sequences, args.cano)
    for i, s in enumerate(instances):
        s["id"] = i
        s["points"] = args.num_points
        if args.synthetic:
            s["synthetic"] = 2 + int(s["seq"].split("_")[1])
    os.makedirs(args.save_root, exist_ok=True)

    if runner is not None:
        def run_and_save(spec):
            res = runner(spec)
            if "seg_part" in res:
                save_instance(args.save_root, spec, res)
            return res

        records, _ = run_sweep(instances, run_and_save, device, policy=args.shard)
    else:
        samples = {}
        make_engine = _engine_factory(args, device, samples)

        def on_finish(inst, spec, eng, en):
            # run_robot.py:333-356 saves result.pkl + model.pth.tar for EVERY optimisation run: without the energy tail (or
            # when it raised for this instance) the files carry the engine's own labels and poses, unmerged
            try:
                if en is None:
                    ident = torch.arange(eng.model.num_parts if hasattr(eng.model, "num_parts") else 0)
                    en = {"seg_part": eng.seg_part, "trans_list": _engine_poses(eng), "joint_connection": ident.new_zeros((0, 2))}
                save_instance(args.save_root, spec, en, samples.get(inst), eng.model, float(eng.tau.item()))
            finally:
                samples.pop(inst, None)

        records, _ = run_sweep_engines(instances, make_engine, args.n_iter, device, per_gpu=args.per_gpu,
                                       chunk=min(100, args.n_iter), energy=args.energy, mode=args.mode, on_finish=on_finish,
                                       policy=args.shard, groups_in_flight=args.groups_in_flight,
                                       assign=dict(assign_iter=args.assign_iter, assign_gap=args.assign_gap, downsample=args.downsample,
                                                   lambda_assign=args.lambda_assign) if args.use_assign_loss else None)
    records = records.cpu()
    win = winners(instances, records)
    owner = owner_of(deal(instances, world, args.shard))
    backend = dist.get_backend() if dist.is_initialized() else None
    # who ran what: every rank reports its device; rank 0 writes the table (one small object gather, outside the data path)
    me = {"rank": rank, "device": str(device), "instances": [i for i, r in owner.items() if r == rank],
          "cano_idx": [instances[i]["cano_idx"] for i, r in owner.items() if r == rank]}
    if dist.is_initialized():
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    else:
        ranks_info = [me]
    if dist.is_initialized():
        dist.barrier()                      # every rank's result files are on disk before rank 0 copies the winners
    if rank == 0:
        fields = ["instance", "cano_idx", "recon_loss", "flow_loss", "total_loss", "iterations", "failed", "parts",
                  "total_err", "ass_err", "screw_err", "group_err", "cd_err"]
        table = {}
        for name in win:
            rows = []
            for i, s in enumerate(instances):
                if s["seq"] != name:
                    continue
                row = {k: (None if bool(torch.isnan(records[i, j])) else float(records[i, j])) for j, k in enumerate(fields)}
                row["rank"] = owner[i]
                rows.append(row)
            w = win[name]
            table[name] = {"winner_cano_idx": None if w is None else instances[w]["cano_idx"],
                           "winner_instance": w, "selected_by": "total_err" if args.energy or runner is not None else "total_loss",
                           "instances": rows}
            if w is not None:
                src = os.path.join(instance_dir(args.save_root, instances[w]), "result.pkl")
                if os.path.exists(src):     # the winner's files, where a single run of the reference would have put them
                    shutil.copyfile(src, os.path.join(args.save_root, name, "result.pkl"))
                    mp = os.path.join(instance_dir(args.save_root, instances[w]), "model.pth.tar")
                    if os.path.exists(mp):
                        shutil.copyfile(mp, os.path.join(args.save_root, name, "model.pth.tar"))
        wall = time.perf_counter() - t_start          # this rank's wall clock from argument parsing to the winners (after the gather)
        rate = len(instances) * args.n_iter / wall
        with open(os.path.join(args.save_root, "sweep.json"), "w") as f:
            json.dump({"world_size": world, "rccl_world": dist.get_world_size() if dist.is_initialized() else 1,
                       "backend": backend, "shard": args.shard, "ranks": ranks_info,
                       "n_instances": len(instances), "n_iter": args.n_iter, "energy": bool(args.energy),
                       "wall_s": round(wall, 3), "iterations_per_s": round(rate, 1),
                       "rank0_stages": getattr(run_sweep_engines, "last_stages", None), "sequences": table}, f, indent=1)
        print(json.dumps({"sweep": os.path.join(args.save_root, "sweep.json"), "n_gpus": world,
                          "rccl_world": dist.get_world_size() if dist.is_initialized() else 1, "backend": backend,
                          "shard": args.shard, "ranks": ranks_info, "instances": len(instances),
                          "wall_s": round(wall, 3), "iterations_per_s": round(rate, 1),
                          "rank0_stages": getattr(run_sweep_engines, "last_stages", None),
                          "winners": {k: v["winner_cano_idx"] for k, v in table.items()}}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
