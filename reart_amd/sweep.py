"""Multi-GPU sweep: independent optimisation instances sharded across ranks.

One optimisation *instance* = (sequence, cano_idx, seed); instances share nothing (the reference's
``main(args)`` is self-contained, run_robot.py:35-358) and the reference selects the canonical frame
"by the lowest energy" (README.md:60) -- so the natural multi-GPU job is the sweep over
``cano_idx`` x sequences.  Ranks own whole instances (static round-robin, no data-path
collective); the only exchange is one ``all_gather`` of a fixed-size float record per instance
(RCCL over xGMI when the backend is "nccl"; latency-bound: 64 B per instance), after which every
rank can take the arg-min.
"""
import torch
import torch.distributed as dist

RECORD = 8  # floats per instance: [instance id, cano_idx, recon, flow, total, iterations, 0, 0]


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
            rec = [inst, spec.get("cano_idx", -1), res["recon"], res["flow"], res["total"], res["iterations"], 0, 0]
        except Exception:  # a failed instance is reported (NaN energy), it does not kill the job
            rec = [inst, spec.get("cano_idx", -1), float("nan"), float("nan"), float("nan"), 0, 1, 0]
        local[inst] = torch.tensor(rec, dtype=torch.float32)
    records = gather_records(local, len(instances), device)
    total = records[:, 4].clone()
    total[torch.isnan(total)] = float("inf")
    return records, int(torch.argmin(total).item())


def run_sweep_engines(instances, make_engine, n_iter, device, per_gpu=3, chunk=100):
    """Sweep of fused-loop instances with ``per_gpu`` of them in flight per GPU.

    One instance of the relaxation loop is a chain of short, latency-bound launches that leaves
    issue slots idle; independent instances on separate streams fill them (measured on MI355X:
    8.0 k it/s for one instance, 14.2 k it/s aggregate for three).  ``make_engine(spec)`` returns a
    prepared ``reart_amd.relax.RelaxEngine`` (its tensors live on ``device``); this rank's instances
    are optimised ``per_gpu`` at a time, stepped round-robin in graph replays of ``chunk``
    iterations.  Returns (records [n, RECORD], index of the lowest-energy instance) like ``run_sweep``."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard(len(instances), rank, world)
    local = {}
    for g0 in range(0, len(mine), per_gpu):
        group = mine[g0:g0 + per_gpu]
        live = []
        for inst in group:
            spec = instances[inst]
            st = torch.cuda.Stream(device=device)
            try:
                with torch.cuda.stream(st):
                    eng = make_engine(spec)
                    done = eng.capture(steps_per_graph=min(chunk, n_iter))
                live.append([inst, spec, eng, st, done])
            except Exception:  # a failed instance is reported (NaN energy), it does not kill the job
                local[inst] = torch.tensor([inst, spec.get("cano_idx", -1), float("nan"), float("nan"), float("nan"), 0, 1, 0],
                                           dtype=torch.float32)
        while any(e[4] < n_iter for e in live):
            for e in live:
                if e[4] < n_iter:
                    n = min(chunk, n_iter - e[4])
                    with torch.cuda.stream(e[3]):
                        e[2].step(n)
                    e[4] += n
        for inst, spec, eng, st, done in live:
            st.synchronize()
            row = eng.last_losses().cpu()
            local[inst] = torch.tensor([inst, spec.get("cano_idx", -1), float(row[0]), float(row[1]), float(row[2]), done, 0, 0],
                                       dtype=torch.float32)
    records = gather_records(local, len(instances), device)
    total = records[:, 4].clone()
    total[torch.isnan(total)] = float("inf")
    return records, int(torch.argmin(total).item())
