"""Multi-GPU sweep: independent optimisation instances sharded across ranks.

One optimisation *instance* = (sequence, cano_idx, seed); instances share nothing (the reference's
``main(args)`` is self-contained, run_robot.py:35-358) and the reference selects the canonical frame
"by the lowest energy" (README.md:60) -- so the natural multi-GPU job is the sweep over
``cano_idx`` x sequences.  Ranks own whole instances (static round-robin, no data-path
collective); the only exchange is one ``all_gather`` of a fixed-size float record per instance
(RCCL over xGMI when the backend is "nccl"; latency-bound: 64 B per instance), after which every
rank can take the arg-min.
"""
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


def shard(n_instances, rank, world):
    """Static round-robin assignment: instance i runs on rank i % world."""
    return list(range(rank, n_instances, world))


def gather_records(local, n_instances, device):
    """local: {instance id: 1-D float tensor [RECORD]} of this rank -> [n_instances, RECORD] on every
    rank, ordered by instance id.  Uses all_gather on padded per-rank blocks (equal message size)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    per = (n_instances + world - 1) // world
    block = torch.full((per, RECORD), float("nan"), dtype=torch.float32, device=device)
    for slot, inst in enumerate(shard(n_instances, rank, world)):
        block[slot] = local[inst].to(device=device, dtype=torch.float32)
    if world > 1:
        blocks = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(blocks, block)
    else:
        blocks = [block]
    out = torch.empty((n_instances, RECORD), dtype=torch.float32, device=device)
    for r in range(world):
        for slot, inst in enumerate(shard(n_instances, r, world)):
            out[inst] = blocks[r][slot]
    return out


def run_sweep(instances, run_instance, device):
    """instances: list of dicts (at least ``cano_idx``); ``run_instance(spec) -> dict(recon, flow,
    total, iterations)`` optimises one instance on this rank's GPU.  Returns (records
    [n, RECORD], index of the lowest-energy instance)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    local = {}
    for inst in shard(len(instances), rank, world):
        spec = instances[inst]
        try:
            res = run_instance(spec)
            local[inst] = _record(inst, spec, (res["recon"], res["flow"], res["total"]), res["iterations"], 0,
                                  res if "total_err" in res else None)
        except Exception:  # a failed instance is reported (NaN energy), it does not kill the job
            local[inst] = _record(inst, spec, failed=1)
    records = gather_records(local, len(instances), device)
    return records, best_instance(records)


def run_sweep_engines(instances, make_engine, n_iter, device, per_gpu=3, chunk=100, energy=False, mode="streams",
                      overlap_tails=True):
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
    Returns (records [n, RECORD], best index) like ``run_sweep``."""
    if mode not in ("streams", "batch"):
        raise ValueError("mode is 'streams' or 'batch'")
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard(len(instances), rank, world)
    local = {}
    threads = os.environ.get("REART_SWEEP_THREADS", "1") != "0"
    pool, pending = None, []
    for g0 in range(0, len(mine), per_gpu):
        group = mine[g0:g0 + per_gpu]
        live = []
        for inst in group:
            spec = instances[inst]
            st = torch.cuda.Stream(device=device)
            try:
                with torch.cuda.stream(st):
                    eng = make_engine(spec)
                    done = eng.capture(steps_per_graph=min(chunk, n_iter)) if mode == "streams" else 0
                live.append([inst, spec, eng, st, done])
            except Exception:  # a failed instance is reported (NaN energy), it does not kill the job
                local[inst] = _record(inst, spec, failed=1)
        if mode == "batch" and live:
            from .relax import RelaxBatch

            for e in live:
                e[3].synchronize()                       # the engines were prepared on their own streams
            for b0 in range(0, len(live), RelaxBatch.MAX):
                part = live[b0:b0 + RelaxBatch.MAX]
                batch = RelaxBatch([e[2] for e in part])
                used = batch.capture(steps_per_graph=min(chunk, n_iter - 1)) if n_iter > 1 else 0
                batch.step(n_iter - used)
                for e in part:
                    e[4] = n_iter
            torch.cuda.current_stream(device).synchronize()
        while any(e[4] < n_iter for e in live):          # streams: round-robin graph replays (nothing left to do after a batch)
            for e in live:
                if e[4] < n_iter:
                    n = min(chunk, n_iter - e[4])
                    with torch.cuda.stream(e[3]):
                        e[2].step(n)
                    e[4] += n
        def finish(entry):
            inst, spec, eng, st, done = entry
            st.synchronize()
            row = eng.last_losses().cpu()
            en = None
            if energy:
                try:
                    with torch.cuda.stream(st):
                        en = instance_energy(eng, spec)
                except Exception:      # e.g. every part merged away: the losses still describe the instance
                    en = None
            return inst, _record(inst, spec, (float(row[0]), float(row[1]), float(row[2])), done, 0, en)

        # The end of an instance is latency-bound (its assignment solves occupy T-1 of the 256 compute units) and full of
        # host round trips: the instances of a group finish side by side, each on its own stream and host thread -- and
        # (overlap_tails) while the NEXT group already optimises: the tails leave nine tenths of the chip idle.
        if energy and threads:
            if pool is None:
                pool = ThreadPoolExecutor(max_workers=2 * max(per_gpu, 1))
            futures = [pool.submit(finish, e) for e in live]
            if overlap_tails:
                pending.extend(futures)
            else:
                for f in futures:
                    inst, rec = f.result()
                    local[inst] = rec
        else:
            for inst, rec in (finish(e) for e in live):
                local[inst] = rec
    for f in pending:
        inst, rec = f.result()
        local[inst] = rec
    if pool is not None:
        pool.shutdown()
    records = gather_records(local, len(instances), device)
    return records, best_instance(records)
