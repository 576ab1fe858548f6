"""CPU: the sweep COMMAND LINE (python -m reart_amd.sweep, BASELINE configs[3]) with two gloo ranks, and the launch helper
bench.py / the sweep use to start one rank per GPU.  The per-instance runner is the ORACLE's relaxation step (tests may
use the oracle); on a GPU node the runner is RelaxEngine and the backend nccl (= RCCL)."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQ_ROOT = os.path.join(ROOT, "tests", "golden")


def _oracle_runner(spec):
    """One instance on the host: the reference loader's sample -> a few oracle iterations -> a made-up structure and an
    energy that depends on the instance (lowest for cano_idx 2)."""
    from oracle.step import RelaxOracle
    from reart_amd.dataset import Sequence

    sample = Sequence(spec["seq_path"], num_points=64, cano_idx=spec["cano_idx"])[0]
    cano, pcs = sample["cano_pc"], sample["pc_list"]
    rng = np.random.default_rng(5)
    H, P, B = 8, 3, pcs.shape[0]
    orc = RelaxOracle(cano, pcs, rng.normal(0, .5, (H, 3)), rng.normal(0, .1, H), rng.normal(0, .2, (P, H)),
                      np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1)), np.zeros((B, P, 3), np.float32),
                      spec["cano_idx"], n_iter=10)
    for _ in range(2):
        out = orc.step(-np.log(rng.exponential(size=(cano.shape[0], P))).astype(np.float32))
    if spec["cano_idx"] == 3:
        raise RuntimeError("injected failure")
    ass = 1.0 + abs(spec["cano_idx"] - 2)
    return dict(recon=out["recon"], flow=0.0, total=out["total"], iterations=2, parts=P, ass_err=ass, screw_err=0.25,
                group_err=0.5, total_err=ass + 0.75, cd_err=0.1,
                seg_part=np.zeros(cano.shape[0], np.int64), trans_list=np.tile(np.eye(4, dtype=np.float32), (B, P, 1, 1)),
                joint_connection=np.array([[0, 1], [1, 2]]))


def _rank(rank, world, port, save_root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from reart_amd import sweep

    rc = sweep.main(["--seq_root", SEQ_ROOT, "--seqs", "seq_tiny", "--cano", "all", "--n_iter", "2", "--energy",
                     "--gpus", str(world), "--save_root", save_root], runner=_oracle_runner)
    assert rc == 0


def test_sweep_cli_two_ranks_gloo(tmp_path):
    from reart_amd.launch import free_port

    ctx = mp.get_context("spawn")
    port = free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    sw = json.load(open(tmp_path / "sweep.json"))
    assert sw["world_size"] == 2 and sw["n_instances"] == 4
    seq = sw["sequences"]["seq_tiny"]
    assert seq["winner_cano_idx"] == 2 and seq["selected_by"] == "total_err"
    rows = seq["instances"]
    assert [r["cano_idx"] for r in rows] == [0, 1, 2, 3] and [r["rank"] for r in rows] == [0, 1, 0, 1]
    assert rows[3]["failed"] == 1 and rows[3]["total_err"] is None          # reported, the job survives
    assert abs(rows[0]["total_err"] - 3.75) < 1e-6 and abs(rows[2]["total_err"] - 1.75) < 1e-6
    # every finished instance left its result.pkl (both ranks wrote), and the winner's sits where one run would put it
    for c in (0, 1, 2):
        assert (tmp_path / "seq_tiny" / f"cano_{c}" / "result.pkl").exists()
    res = pickle.load(open(tmp_path / "seq_tiny" / "result.pkl", "rb"))
    assert res["cano_idx"] == 2 and set(res) >= {"pred_cano_part", "pred_pose_list", "cano_idx", "joint_connection"}
    # the same instance in a single process gives the same record
    single = _oracle_runner(dict(seq_path=os.path.join(SEQ_ROOT, "seq_tiny"), cano_idx=0))
    assert abs(single["total"] - rows[0]["total_loss"]) <= 1e-5 * abs(single["total"])


def test_enumeration_and_winners():
    import torch

    from reart_amd import sweep

    seqs = sweep.list_sequences(SEQ_ROOT)
    assert seqs == [("seq_tiny", os.path.join(SEQ_ROOT, "seq_tiny"), 4)]
    inst = sweep.enumerate_instances([("a", "/a", 3), ("b", "/b", 2)], "all")
    assert [(s["seq"], s["cano_idx"]) for s in inst] == [("a", 0), ("a", 1), ("a", 2), ("b", 0), ("b", 1)]
    assert [s["cano_idx"] for s in sweep.enumerate_instances([("a", "/a", 5)], "1,3")] == [1, 3]
    try:
        sweep.enumerate_instances([("a", "/a", 2)], "2")
        raise AssertionError("cano_idx outside the sequence must be refused")
    except ValueError:
        pass
    rec = torch.full((5, sweep.RECORD), float("nan"))
    rec[:, 4] = torch.tensor([3.0, 1.0, 2.0, 5.0, 4.0])                  # losses only: they decide
    assert sweep.winners(inst, rec) == {"a": 1, "b": 4}
    rec[:, sweep.E_TOTAL] = torch.tensor([0.3, 0.9, float("nan"), float("nan"), float("nan")])
    assert sweep.winners(inst, rec) == {"a": 0, "b": 4}                  # energies where there are any
    rec[3:, 4] = float("nan")
    assert sweep.winners(inst, rec)["b"] is None


def test_launch_helper():
    from reart_amd import launch

    cmd = launch.torchrun_command("bench.py", ["--gpus", "4"], 4, port=1234)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-3:] == ["bench.py", "--gpus", "4"]
    assert launch.torchrun_command("reart_amd.sweep", [], 2, port=1, module=True)[-2:] == ["-m", "reart_amd.sweep"]
    assert launch.check_world(1, {}) == 1 and launch.check_world(8, {"WORLD_SIZE": "8"}) == 8
    try:
        launch.check_world(8, {"WORLD_SIZE": "1"})
        raise AssertionError("a 1-rank job must not pass for --gpus 8")
    except SystemExit:
        pass
    assert launch.under_launcher({"RANK": "0", "WORLD_SIZE": "2"}) and not launch.under_launcher({})


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    """`bench.py --gpus 2` inside a ONE-rank job must fail, not print an n_gpus = 1 line."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, timeout=300)
    assert r.returncode != 0 and b"--gpus 2" in r.stderr and b"n_gpus" not in r.stdout


def _rank4(rank, world, port, seq_root, save_root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from reart_amd import sweep

    rc = sweep.main(["--seq_root", seq_root, "--cano", "all", "--n_iter", "2", "--energy", "--gpus", str(world), "--shard", "lpt",
                     "--save_root", save_root], runner=_oracle_runner)
    assert rc == 0


def test_sweep_cli_four_ranks_lpt_with_unequal_sequences(tmp_path):
    """world 4 (gloo), two sequences of different length (4 and 2 frames: instance costs 2 : 1 by frames x points^2), dealt
    longest first (--shard lpt): the deal is the documented one, every rank's records arrive under the right instance id
    (unequal per-rank counts: padded blocks), each sequence gets its own winner, and the line says who ran what."""
    import shutil

    from reart_amd import sweep
    from reart_amd.launch import free_port

    root = tmp_path / "seqs"
    shutil.copytree(os.path.join(SEQ_ROOT, "seq_tiny"), root / "seq_a")
    (root / "seq_b").mkdir()
    for f in ("state_0.pkl", "state_1.pkl", "pose_1.pkl"):            # a 2-frame sequence in the reference's layout
        shutil.copy(os.path.join(SEQ_ROOT, "seq_tiny", f), root / "seq_b" / f)
    seqs = sweep.list_sequences(str(root))
    assert [(n, t) for n, _, t in seqs] == [("seq_a", 4), ("seq_b", 2)]
    inst = sweep.enumerate_instances(seqs, "all")
    for s in inst:
        s["points"] = 4096
    plan = sweep.deal(inst, 4, "lpt")
    assert plan == [[0, 4], [1, 5], [2], [3]]                          # the four long instances first, one per rank; the short ones fill up
    assert sweep.deal(inst, 4, "round_robin") == [[0, 4], [1, 5], [2], [3]]
    assert sweep.deal(list(reversed(inst)), 4, "lpt") == [[2, 0], [3, 1], [4], [5]]   # order of enumeration does not matter to LPT
    out = tmp_path / "out"
    ctx = mp.get_context("spawn")
    port = free_port()
    procs = [ctx.Process(target=_rank4, args=(r, 4, port, str(root), str(out))) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    sw = json.load(open(out / "sweep.json"))
    assert sw["world_size"] == 4 and sw["rccl_world"] == 4 and sw["backend"] == "gloo" and sw["shard"] == "lpt"
    assert sw["n_instances"] == 6 and [r["instances"] for r in sw["ranks"]] == plan
    a, b = sw["sequences"]["seq_a"]["instances"], sw["sequences"]["seq_b"]["instances"]
    assert [r["rank"] for r in a] == [0, 1, 2, 3] and [r["rank"] for r in b] == [0, 1]
    assert [r["cano_idx"] for r in a] == [0, 1, 2, 3] and [r["cano_idx"] for r in b] == [0, 1]
    assert a[3]["failed"] == 1 and all(r["failed"] == 0 for r in a[:3] + b)        # the injected failure is reported, nothing else lost
    assert sw["sequences"]["seq_a"]["winner_cano_idx"] == 2 and sw["sequences"]["seq_b"]["winner_cano_idx"] == 1
