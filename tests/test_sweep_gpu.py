"""Concurrent instances per GPU (sweep.run_sweep_engines): every instance must end exactly where the same
instance ends when it runs alone -- streams only interleave the launches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["streams", "batch"])
def test_concurrent_instances_match_solo_runs(dev, mode):
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.sweep import run_sweep_engines
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=5, n_parts=3, pts_per_part=200, seed=4, n_ref=300, with_flow=True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def make_engine(spec):
        cano, pcs = split_canonical(seq["complete"], spec["cano_idx"])
        torch.manual_seed(spec["cano_idx"])
        model = BaseModel(num_parts=6, pose_len=4).to(dev)
        return RelaxEngine(t(cano), t(pcs), model, spec["cano_idx"], [t(r) for r in seq["ref_loc"]],
                           [t(f) for f in seq["ref_flow"]], n_iter=60, seed=7 + spec["cano_idx"])

    instances = [{"cano_idx": c} for c in range(5)]
    rec, best = run_sweep_engines(instances, make_engine, 60, dev, per_gpu=3, chunk=20, mode=mode)
    rec = rec.cpu().numpy()
    assert np.isfinite(rec[:, 2:5]).all() and (rec[:, 5] == 60).all()
    for c in range(5):
        eng = make_engine({"cano_idx": c})
        eng.step(60)
        solo = eng.last_losses().cpu().numpy()
        np.testing.assert_array_equal(rec[c, 2:5], solo[:3])
    assert best == int(np.argmin(rec[:, 4]))


def test_sweep_records_carry_the_reference_energy(dev):
    """energy=True: every instance ends with structure extraction + the model-selection energy (run_robot.py:306-321);
    the record equals a solo run's tail and the winner is the lowest total energy."""
    from reart_amd import sweep
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=5, n_parts=3, pts_per_part=300, seed=4, with_flow=False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def make_engine(spec):
        cano, pcs = split_canonical(seq["complete"], spec["cano_idx"])
        torch.manual_seed(spec["cano_idx"])
        model = BaseModel(num_parts=6, pose_len=4).to(dev)
        return RelaxEngine(t(cano), t(pcs), model, spec["cano_idx"], n_iter=400, seed=3 + spec["cano_idx"])

    instances = [{"cano_idx": c} for c in (1, 2, 3)]
    rec, best = sweep.run_sweep_engines(instances, make_engine, 400, dev, per_gpu=3, chunk=100, energy=True)
    rec = rec.cpu().numpy()
    assert rec.shape == (3, sweep.RECORD) and np.isfinite(rec[:, 7:13]).all()
    np.testing.assert_allclose(rec[:, 8], rec[:, 9] + rec[:, 10] + rec[:, 11], rtol=1e-6)
    assert best == int(np.argmin(rec[:, 8]))
    eng = make_engine(instances[1])
    eng.step(400)
    solo = sweep.instance_energy(eng, instances[1])
    np.testing.assert_allclose(rec[1, 8:12], [solo["total_err"], solo["ass_err"], solo["screw_err"], solo["group_err"]], rtol=1e-6)
    assert rec[1, 7] == solo["parts"]
    # the energy is computed on the CALLER's point order (the engine stores k-d order; the tail depends on point order):
    # it is the value the stand-alone tail gives on the original clouds
    from reart_amd import tail

    cano, pcs = split_canonical(seq["complete"], 2)
    direct = tail.finish_instance(eng.model, t(cano), t(pcs), 2)
    for k in ("total_err", "ass_err", "screw_err", "group_err"):
        assert direct[k] == solo[k], k
    c2, p2 = eng.caller_clouds()
    np.testing.assert_array_equal(c2.cpu().numpy(), cano)
    np.testing.assert_array_equal(p2.cpu().numpy(), pcs)


def test_sweep_command_line_one_rank(dev, tmp_path):
    """python -m reart_amd.sweep on a sequence directory in the reference's layout (tests/golden/seq_tiny, 4 frames): every
    cano_idx is optimised by the fused engine, ends with the reference's energy, and the lowest total energy wins
    (README.md:58-60); result files carry the reference's keys (run_robot.py:333-356)."""
    import json
    import os
    import pickle

    from reart_amd import sweep

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rc = sweep.main(["--seq_root", root, "--seqs", "seq_tiny", "--cano", "all", "--n_iter", "300", "--energy", "--num_points", "80",
                     "--num_parts", "6", "--per_gpu", "2", "--save_root", str(tmp_path)])
    assert rc == 0
    sw = json.load(open(tmp_path / "sweep.json"))
    seq = sw["sequences"]["seq_tiny"]
    rows = seq["instances"]
    assert sw["world_size"] == 1 and [r["cano_idx"] for r in rows] == [0, 1, 2, 3]
    assert all(r["iterations"] == 300 and r["failed"] == 0 and np.isfinite(r["total_loss"]) for r in rows)
    en = [r["total_err"] for r in rows]
    assert all(e is not None for e in en), en
    w = int(np.argmin(en))
    assert seq["winner_cano_idx"] == w and seq["selected_by"] == "total_err"
    for r in rows:
        assert abs(r["total_err"] - (r["ass_err"] + r["screw_err"] + r["group_err"])) <= 1e-5 * abs(r["total_err"])
    res = pickle.load(open(tmp_path / "seq_tiny" / "result.pkl", "rb"))
    assert res["cano_idx"] == w and res["pred_cano_part"].shape == (80,) and res["pred_pose_list"].shape[0] == 3
    assert set(res) >= {"pred_cano_part", "pred_pose_list", "cano_idx", "joint_connection", "cano_pc", "pc_list"}
    ck = torch.load(tmp_path / "seq_tiny" / "model.pth.tar", weights_only=False)
    assert set(ck) >= {"state_dict", "tau", "cano_idx"} and ck["cano_idx"] == w
    # a solo run of the winner ends with the same energy the sweep recorded
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd import run_robot as rr

    sample = rr.load_sequence(os.path.join(root, "seq_tiny"), 80, w)
    torch.manual_seed(2)
    model = BaseModel(num_parts=6, pose_len=3).to(dev)
    eng = RelaxEngine(torch.from_numpy(sample["cano_pc"]).float().to(dev), torch.from_numpy(sample["pc_list"]).float().to(dev),
                      model, w, n_iter=300, seed=2)
    eng.step(300)
    solo = sweep.instance_energy(eng, {"cano_idx": w})
    assert abs(solo["total_err"] - rows[w]["total_err"]) <= 1e-6 * abs(solo["total_err"])


def test_sweep_command_line_synthetic_batch_with_flow(dev, tmp_path):
    """Two generated sequences x all canonical frames, flow loss on, the default shared-launch mode (groups of six, the last
    group smaller): every instance finishes, one winner per sequence."""
    import json

    from reart_amd import sweep

    rc = sweep.main(["--synthetic", "2", "--synthetic_frames", "5", "--num_points", "512", "--cano", "all", "--n_iter", "150",
                     "--use_flow_loss", "--energy", "--save_root", str(tmp_path)])
    assert rc == 0
    sw = json.load(open(tmp_path / "sweep.json"))
    assert sw["n_instances"] == 10 and set(sw["sequences"]) == {"synthetic_0", "synthetic_1"}
    for name, seq in sw["sequences"].items():
        rows = seq["instances"]
        assert [r["cano_idx"] for r in rows] == [0, 1, 2, 3, 4]
        assert all(r["iterations"] == 150 and r["failed"] == 0 and np.isfinite(r["total_loss"]) and r["flow_loss"] > 0 for r in rows)
        have = [r for r in rows if r["total_err"] is not None]
        assert have, "no instance of the sequence produced an energy"
        assert seq["winner_cano_idx"] == min(have, key=lambda r: r["total_err"])["cano_idx"]
        assert (tmp_path / name / "result.pkl").exists()


@pytest.mark.parametrize("per_gpu,in_flight", [(None, None), (2, None), (2, 2), (1, 2)])
def test_sweep_command_line_readme_recipe(dev, tmp_path, per_gpu, in_flight):
    """in_flight 2: more groups (3, or 5 of one instance) than may run at once -- the later groups are built and
    captured lazily by the worker whose group has finished; never more than two in flight, same results.
    per_gpu 2: the five instances run as THREE groups (2 + 2 + 1) that go through both phases CONCURRENTLY, each on a
    stream and a host thread of its own (round 5) -- same results, instance by instance, as one group of five.
    The README recipe as a sweep (README.md:116: --use_flow_loss --use_assign_loss --downsample 4; here assign_iter 40 of
    80 iterations): the five canonical frames of a generated sequence step in SHARED launches through both phases -- no
    fall-back to streams --, every refresh solves the 5 x 4 assignment problems of 512 x 512 in one call, nothing goes to the
    host solver, and an instance ends exactly as a solo engine driven by AssignmentPhase ends."""
    import json
    import warnings

    from reart_amd import run_robot as rr
    from reart_amd import sweep
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine

    argv = ["--synthetic", "1", "--synthetic_frames", "5", "--num_points", "2048", "--cano", "all", "--n_iter", "80", "--use_flow_loss",
            "--use_assign_loss", "--assign_iter", "40", "--assign_gap", "5", "--downsample", "4", "--save_root", str(tmp_path)]
    if per_gpu:
        argv += ["--per_gpu", str(per_gpu)]
    if in_flight:
        argv += ["--groups_in_flight", str(in_flight)]
    with warnings.catch_warnings():
        warnings.simplefilter("error")                  # a batch group falling back to streams warns: not allowed here
        assert sweep.main(argv) == 0
    sw = json.load(open(tmp_path / "sweep.json"))
    st = sw["rank0_stages"]
    groups = {None: 1, 2: 3, 1: 5}[per_gpu]
    assert st["assign_refreshes"] == 8 * groups and st["lap_fallbacks"] == 0         # per group: refreshes at 40, 45, ..., 75
    assert st.get("concurrent_groups", 1) == min(groups, in_flight or 4)
    assert st.get("groups_in_flight_max", 1) <= (in_flight or 4)
    rows = sw["sequences"]["synthetic_0"]["instances"]
    assert all(r["iterations"] == 80 and r["failed"] == 0 and np.isfinite(r["total_loss"]) for r in rows)
    # instance cano_idx 1, alone
    args = sweep.build_cli().parse_args(argv)
    sample = rr.synthetic_sequence(2048, 1, 5, True, seed=2)
    cano, pcs = torch.from_numpy(sample["cano_pc"]).float().to(dev), torch.from_numpy(sample["pc_list"]).float().to(dev)
    refs, flows = rr.flow_references(args, sample, dev, None)
    torch.manual_seed(2)
    model = BaseModel(num_parts=20, pose_len=4).to(dev)
    eng = RelaxEngine(cano, pcs, model, 1, refs, flows, n_iter=80, seed=2)
    eng.step(40)
    ph = rr.AssignmentPhase(eng, cano, pcs, 4, 5, 0.3)
    ph.run(40, 80)
    row = eng.last_losses().cpu().numpy()
    assert ph.fallbacks == 0
    np.testing.assert_allclose([rows[1]["recon_loss"], rows[1]["flow_loss"], rows[1]["total_loss"]], row[:3], rtol=1e-6)
