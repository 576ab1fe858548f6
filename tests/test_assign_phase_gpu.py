"""The base model's assignment-loss phase (run_robot.py:164-187) of the README recipe (README.md:116: --use_flow_loss
--use_assign_loss --downsample 4, assign_gap 5) at the reference's demo size, against an ORACLE-side loop: oracle forward /
FPS / cdist, scipy's linear_sum_assignment as the reference calls it, the oracle's iteration with those pairs.  Nothing of
the product is on the checker's side."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_base_recipe_assignment_phase_against_the_oracle_loop(oracle, dev):
    """nao (9 x 4096 points, cano_idx 2) from the reference's shipped base-2 state: 11 iterations of the assignment phase
    with refreshes at iterations 0, 5 and 10 -- `AssignmentPhase.refresh` (production: in-kernel Gumbel noise, GPU FPS,
    cold raced auction then per-wave re-solves, device-side pairs) + `RelaxEngine.step`, against RelaxOracle with the
    exported noise stream.  Sampled indices and the optimal permutations equal, no host fallback, losses 1e-5, clouds 5e-7."""
    from oracle.step import RelaxOracle, tau_cosine
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine, gumbel_noise
    from reart_amd.run_robot import AssignmentPhase

    s, w = np.load(os.path.join(G, "structure.npz")), np.load(os.path.join(G, "base_model.npz"))
    cano, pcs, c = s["cano"], s["pc_list"], int(s["cano_idx"])
    B, N, P, ds, gap, lam, n_iter, seed = pcs.shape[0], pcs.shape[1], 20, 4, 5, 0.3, 15000, 5
    rng = np.random.default_rng(0)
    sel = [rng.permutation(N)[:3000] for _ in range(B)]
    refs = [s["complete_gt_pc_list"][f][x] for f, x in enumerate(sel)]
    flows = [s["gt_flow_list"][f][x] for f, x in enumerate(sel)]
    orc = RelaxOracle(cano, pcs, w["W1"], w["b1"], w["W2"], w["p6d"], w["pt"], c, refs, flows, n_iter=n_iter)
    model = BaseModel(num_parts=P, pose_len=B).to(dev)
    with torch.no_grad():
        model.seg_head.model[0].weight.copy_(t(w["W1"], dev)[:, :, None])
        model.seg_head.model[0].bias.copy_(t(w["b1"], dev))
        model.seg_head.model[2].weight.copy_(t(w["W2"], dev)[:, :, None])
        model.proposal_6d.copy_(t(w["p6d"], dev))
        model.proposal_t.copy_(t(w["pt"], dev))
    eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, c, [t(r, dev) for r in refs], [t(f, dev) for f in flows], n_iter=n_iter, seed=seed)
    phase = AssignmentPhase(eng, t(cano, dev), t(pcs, dev), ds, gap, lam)
    n = N // ds
    # run_robot.py:167-169: both sides sampled by FPS (the CUDA kernel's rules, start 0)
    src_o = oracle.fps(cano[None], n, start=np.zeros(1, np.int64), cuda_mode=True)[0]
    tgt_o = oracle.fps(pcs, n, start=np.zeros(B, np.int64), cuda_mode=True)
    np.testing.assert_array_equal(phase.src_idx.cpu().numpy()[0], src_o)
    # ... the targets then re-numbered along a Z-order curve (lap.spatial_order): a permutation of the reference's sample
    order = phase.tgt_order.cpu().numpy()
    np.testing.assert_array_equal(np.sort(order, axis=1), np.tile(np.arange(n), (B, 1)))
    np.testing.assert_array_equal(phase.tgt_idx.cpu().numpy(), np.take_along_axis(tgt_o, order, axis=1))
    tgt_pts_o = np.stack([pcs[b][tgt_o[b]] for b in range(B)])
    assign = None
    for i in range(2 * gap + 1):
        stored = gumbel_noise(seed, i, N, P, dev)                       # the noise the production forward draws for iteration i
        noise = torch.empty_like(stored)
        noise[eng._perm] = stored                                        # rows in the caller's point order
        noise = noise.cpu().numpy()
        if i % gap == 0:
            p = orc.params
            tau = float(np.float32(tau_cosine(orc.it + 1, n_iter, 1.0, 5.0)))
            X = oracle.base_forward(orc.cano, p["W1"], p["b1"], p["W2"], p["p6d"], p["pt"], noise, tau)["out"]
            cost = oracle.cdist(np.ascontiguousarray(X[:, src_o]), tgt_pts_o)
            cols_o = np.stack([cc for _, cc in oracle.linear_sum_assignment(cost)])          # scipy, like the reference
            assign = (src_o, np.take_along_axis(tgt_o, cols_o, axis=1), lam)
            phase.refresh()
            # (the solver's columns are in the phase's numbering: back to the reference's sample order)
            np.testing.assert_array_equal(np.take_along_axis(order, phase.lap_state["cols"].cpu().numpy().astype(np.int64), axis=1), cols_o,
                                          err_msg=f"refresh at iteration {i}")
            assert phase.fallbacks == 0
        ref = orc.step(noise, assign=assign)
        eng.step()
        row = eng.last_losses().cpu().numpy()
        assert abs(row[0] - ref["recon"]) <= 1e-5 * abs(ref["recon"]), (i, row, ref["recon"])
        assert abs(row[1] - ref["flow"]) <= 1e-5 * abs(ref["flow"]) + 1e-9, (i, row, ref["flow"])
        np.testing.assert_allclose(eng.pc_trans.cpu().numpy(), ref["pc_trans"], rtol=0, atol=5e-7, err_msg=f"iteration {i}")
    assert phase.refreshes == 3 and phase.fallbacks == 0
    assert phase._device_path() and phase._slot is not None        # refreshes 2 and 3 ran without host-side tensor operations


def test_refresh_glue_entry_points(dev):
    """reart_gather_points / reart_assign_pairs against the tensor operations they replace (index_points of the sampled
    source points, run_robot.py:169; the pairs RelaxEngine.set_assignment writes, :177-178), through the C ABI."""
    from reart_amd import _lib
    from reart_amd.networks.pointnet2_utils import index_points

    L, g = _lib.lib(), torch.Generator(device="cpu").manual_seed(3)
    B, N, n = 5, 777, 130
    pc = torch.randn((B, N, 3), generator=g).to(dev)
    idx = torch.randperm(N, generator=g)[:n].to(dev)
    out = torch.full((B, n, 3), float("nan"), device=dev)
    idx32 = idx.int()                                                  # (kept alive: a temporary's memory is reused at once)
    _lib.check(L.reart_gather_points(_lib.ptr(pc), _lib.ptr(idx32), B, N, n, _lib.ptr(out), _lib.stream()), "reart_gather_points")
    assert torch.equal(out, index_points(pc, idx.expand(B, n)))
    cols = torch.stack([torch.randperm(n, generator=g) for _ in range(B)]).to(dev)
    tgt_index = torch.stack([torch.randperm(N, generator=g)[:n] for _ in range(B)]).to(dev)
    slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
    slot[idx] = torch.arange(n, dtype=torch.int32, device=dev)
    amap = torch.full((B, N), 12345, dtype=torch.int32, device=dev)
    cols32, tgt32 = cols.int(), tgt_index.int()
    _lib.check(L.reart_assign_pairs(_lib.ptr(cols32), _lib.ptr(slot), _lib.ptr(tgt32), B, N, n, _lib.ptr(amap), _lib.stream()),
               "reart_assign_pairs")
    want = torch.full((B, N), -1, dtype=torch.int32, device=dev)
    want[:, idx] = tgt_index.gather(1, cols).int()
    assert torch.equal(amap, want)
    assert L.reart_gather_points(None, _lib.ptr(idx32), B, N, n, _lib.ptr(out), _lib.stream()) == -1
    assert L.reart_assign_pairs(_lib.ptr(cols32), _lib.ptr(slot), _lib.ptr(tgt32), B, N, 0, _lib.ptr(amap), _lib.stream()) == -1
    # reart_publish_words: three device arrays -> one pinned host buffer, in order, by one launch (no copy launches); b / c may be empty
    a, b_, c = (torch.arange(k0, k0 + m, dtype=torch.int32, device=dev) for k0, m in ((10, 9), (100, 9), (1000, 36)))
    host = torch.full((9 + 9 + 36 + 3,), -7, dtype=torch.int32).pin_memory()
    _lib.check(L.reart_publish_words(_lib.ptr(a), 9, _lib.ptr(b_), 9, _lib.ptr(c), 36, _lib.c_void_p(host.data_ptr()), _lib.stream()),
               "reart_publish_words")
    torch.cuda.synchronize()
    assert host.tolist() == list(range(10, 19)) + list(range(100, 109)) + list(range(1000, 1036)) + [-7] * 3
    _lib.check(L.reart_publish_words(_lib.ptr(c), 36, None, 0, None, 0, _lib.c_void_p(host.data_ptr()), _lib.stream()), "reart_publish_words")
    torch.cuda.synchronize()
    assert host[:36].tolist() == list(range(1000, 1036))
    assert L.reart_publish_words(None, 3, None, 0, None, 0, _lib.c_void_p(host.data_ptr()), _lib.stream()) == -1
    assert L.reart_publish_words(_lib.ptr(a), 9, None, 0, None, 0, None, _lib.stream()) == -1


def test_device_side_refresh_equals_the_tensor_path(dev):
    """The same 16 iterations of the assignment phase (refreshes at 0, 5, 10, 15) with the refresh glue on the device and with
    the tensor operations it replaces: the optimum is unique, so assignments, pair maps and the trajectory are identical."""
    from reart_amd.data import load_nao_demo
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.run_robot import AssignmentPhase

    g = load_nao_demo()
    cano, pcs, c = t(g["cano"], dev), t(g["pc_list"], dev), int(g["cano_idx"])
    outs = []
    for native in (True, False):
        torch.manual_seed(4)
        model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
        eng = RelaxEngine(cano, pcs, model, c, None, None, n_iter=15000, seed=4)
        eng.step(40)
        phase = AssignmentPhase(eng, cano, pcs, 4, 5, 0.3)
        phase.NATIVE = native
        i = phase.run(40, 56)
        assert i == 56 and phase.refreshes == 4 and phase.fallbacks == 0
        assert (getattr(phase, "_slot", None) is not None) == native
        outs.append((phase.lap_state["cols"].clone(), eng._assign_map.clone(), eng.pc_trans.clone(), eng.last_losses().clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_refresh_schedule_when_the_phase_starts_one_iteration_before_a_refresh(dev):
    """run_robot.py:165 refreshes the pairs at the first assignment iteration AND at every i % assign_gap == 0.  With
    assign_iter % assign_gap == assign_gap - 1 (here 4 and 5) the second refresh is due one iteration after the first -- the
    iteration the graph capture runs eagerly.  `AssignmentPhase.run` (graph replays between refreshes) against the
    reference's loop structure spelled out one iteration at a time: same refreshes (4, 5, 10, 15), same snapshots, the same
    trajectory bit for bit; the shared-launch form (`AssignmentPhaseBatch`) keeps the same schedule."""
    from reart_amd.data import load_nao_demo
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxBatch, RelaxEngine
    from reart_amd.run_robot import AssignmentPhase, AssignmentPhaseBatch

    g = load_nao_demo()
    cano, pcs, c = t(g["cano"], dev), t(g["pc_list"], dev), int(g["cano_idx"])
    start, end, gap = 4, 17, 5

    def engine():
        torch.manual_seed(4)
        model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
        eng = RelaxEngine(cano, pcs, model, c, None, None, n_iter=15000, seed=4)
        eng.step(start)
        return eng

    # the reference's structure, literally
    eng = engine()
    ref = AssignmentPhase(eng, cano, pcs, 4, gap, 0.3)
    refreshed = []
    for i in range(start, end):
        if not ref._have or i % gap == 0:
            ref.refresh()
            refreshed.append(i)
        eng.step(1)
    assert refreshed == [4, 5, 10, 15]
    want = (ref.lap_state["cols"].clone(), eng._assign_map.clone(), eng.pc_trans.clone(), eng.last_losses().clone())
    # the production loop
    eng = engine()
    phase = AssignmentPhase(eng, cano, pcs, 4, gap, 0.3)
    snaps = []
    assert phase.run(start, end, snapshot_gap=5, on_snapshot=snaps.append) == end
    assert phase.refreshes == 4 and phase.fallbacks == 0
    assert snaps == [5, 10, 15, 17]
    for a, b in zip(want, (phase.lap_state["cols"], eng._assign_map, eng.pc_trans, eng.last_losses())):
        assert torch.equal(a, b)
    # two instances in shared launches: the first of them is the instance above
    engs = [engine(), engine()]
    batch = RelaxBatch(engs)
    pb = AssignmentPhaseBatch(batch, [(cano, pcs), (cano, pcs)], 4, gap, 0.3)
    assert pb.run(start, end) == end and pb.refreshes == 4 and pb.fallbacks == 0
    for e in engs:
        assert torch.equal(e.pc_trans, want[2]) and torch.equal(e._assign_map, want[1])


def test_a_batch_graph_captured_in_the_chamfer_phase_is_not_replayed_in_the_assignment_phase(dev):
    """RelaxBatch keeps ONE graph for all its engines; when the engines switch to the assignment loss (set_assignment drops
    their own graphs) the batch's Chamfer-phase graph must go too -- with assign_gap 1 nothing re-captures, and replaying it
    would silently run Chamfer iterations.  Two instances, assign_gap 1: equal to the solo loop."""
    from reart_amd.data import load_nao_demo
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxBatch, RelaxEngine
    from reart_amd.run_robot import AssignmentPhase, AssignmentPhaseBatch

    g = load_nao_demo()
    cano, pcs, c = t(g["cano"], dev), t(g["pc_list"], dev), int(g["cano_idx"])

    def engine():
        torch.manual_seed(4)
        model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
        return RelaxEngine(cano, pcs, model, c, None, None, n_iter=15000, seed=4)

    solo = engine()
    solo.step(3)
    ph = AssignmentPhase(solo, cano, pcs, 4, 1, 0.3)
    ph.run(3, 7)
    engs = [engine(), engine()]
    batch = RelaxBatch(engs)
    done = batch.capture(steps_per_graph=1)                 # the Chamfer phase's graph
    batch.step(3 - done)
    pb = AssignmentPhaseBatch(batch, [(cano, pcs), (cano, pcs)], 4, 1, 0.3)
    pb.run(3, 7)
    assert pb.refreshes == ph.refreshes == 4
    for e in engs:
        assert torch.equal(e.pc_trans, solo.pc_trans) and torch.equal(e.last_losses(), solo.last_losses())
