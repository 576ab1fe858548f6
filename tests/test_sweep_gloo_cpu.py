"""CPU, world_size 2, gloo: the N > 1 path of the instance sweep (sharding + gather of energies).
The per-instance runner here is the ORACLE's relaxation step on a tiny problem (tests may use the
oracle); on a GPU node the runner is RelaxEngine and the backend is nccl (= RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _runner(spec):
    from oracle.step import RelaxOracle
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=4, n_parts=2, pts_per_part=32, seed=7, with_flow=False)
    cano, pcs = split_canonical(seq["complete"], spec["cano_idx"])
    rng = np.random.default_rng(0)
    H, P, B = 16, 4, 3
    orc = RelaxOracle(cano, pcs, rng.normal(0, .5, (H, 3)), rng.normal(0, .1, H), rng.normal(0, .2, (P, H)),
                      np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1)), np.zeros((B, P, 3), np.float32),
                      spec["cano_idx"], n_iter=10)
    out = None
    for _ in range(3):
        out = orc.step(-np.log(rng.exponential(size=(cano.shape[0], P))).astype(np.float32))
    if spec.get("fail"):
        raise RuntimeError("injected failure")
    res = dict(recon=out["recon"], flow=0.0, total=out["total"], iterations=3)
    if spec.get("with_energy"):   # the end-of-run energy terms travel in the same record (run_robot.py:306-321)
        res.update(parts=4, ass_err=0.1 * (1 + spec["cano_idx"]), screw_err=0.01, group_err=0.02, cd_err=0.5)
        res["total_err"] = res["ass_err"] + res["screw_err"] + res["group_err"]
    return res


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reart_amd.sweep import run_sweep, shard

    instances = [dict(cano_idx=i % 4, fail=(i == 3)) for i in range(5)]
    assert shard(5, rank, world) == list(range(rank, 5, world))
    rec, best = run_sweep(instances, _runner, torch.device("cpu"))
    q.put((rank, rec.numpy(), best))
    dist.destroy_process_group()


def test_sweep_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, rec0, best0), (r1, rec1, best1) = sorted(got, key=lambda x: x[0])
    np.testing.assert_array_equal(rec0, rec1)            # every rank holds the same gathered table
    assert best0 == best1
    assert list(rec0[:, 0]) == [0, 1, 2, 3, 4]           # ordered by instance id
    assert np.isnan(rec0[3, 4]) and rec0[3, 6] == 1      # failed instance reported, job survives
    ok = [0, 1, 2, 4]
    assert np.isfinite(rec0[ok, 4]).all() and best0 == ok[int(np.argmin(rec0[ok, 4]))]
    # same instance on either rank gives the same energy as a single-process run
    single = _runner(dict(cano_idx=0))
    assert abs(single["total"] - rec0[0, 4]) <= 1e-5 * abs(single["total"])


def test_records_with_energy_pick_the_lowest_energy():
    """Single process: 16-float records, energies decide the winner (README.md:60), NaN-energy instances lose."""
    from reart_amd import sweep

    instances = [dict(cano_idx=2, with_energy=True), dict(cano_idx=0, with_energy=True), dict(cano_idx=1, fail=True),
                 dict(cano_idx=3, with_energy=True)]
    rec, best = sweep.run_sweep(instances, _runner, torch.device("cpu"))
    rec = rec.numpy()
    assert rec.shape == (4, sweep.RECORD) and best == 1
    np.testing.assert_allclose(rec[[0, 1, 3], sweep.E_TOTAL], [0.33, 0.13, 0.43], rtol=1e-6)
    np.testing.assert_allclose(rec[1, 7:13], [4, 0.13, 0.1, 0.01, 0.02, 0.5], rtol=1e-6)
    assert np.isnan(rec[2, sweep.E_TOTAL]) and rec[2, 6] == 1
