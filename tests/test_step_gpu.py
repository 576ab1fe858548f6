"""GPU parity of the fused relaxation iteration (reart_relax_step) against the oracle's
iteration and against the reference's golden loss trajectory."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _make_model(dev, P, B, W1, b1, W2, p6d=None, pt=None):
    from reart_amd.networks.model import BaseModel

    m = BaseModel(num_parts=P, pose_len=B).to(dev)
    with torch.no_grad():
        m.seg_head.model[0].weight.copy_(t(W1, dev)[:, :, None])
        m.seg_head.model[0].bias.copy_(t(b1, dev))
        m.seg_head.model[2].weight.copy_(t(W2, dev)[:, :, None])
        if p6d is not None:
            m.proposal_6d.copy_(t(p6d, dev))
        if pt is not None:
            m.proposal_t.copy_(t(pt, dev))
    return m


def test_fused_step_reproduces_reference_trajectory(dev):
    """G11: 10 iterations of the reference loop (BaseModel + recon_loss + torch Adam) with the
    captured Gumbel noise; fp32 losses within 1e-4 relative (BASELINE north_star tolerance)."""
    from reart_amd.relax import RelaxEngine

    g = np.load(os.path.join(G, "trajectory.npz"))
    model = _make_model(dev, 20, 9, g["W1_0"], g["b1_0"], g["W2_0"])
    eng = RelaxEngine(t(g["cano"], dev), t(g["pcs"], dev), model, cano_idx=2, n_iter=15000)
    for i in range(10):
        eng.set_gumbel(t(g["noises"][i], dev))
        eng.step()
        row = eng.last_losses().cpu().numpy()
        assert abs(row[0] - g["losses"][i]) <= 1e-4 * g["losses"][i], (i, row, g["losses"][i])
        assert abs(row[3] - g["taus"][i]) < 1e-6
    np.testing.assert_allclose(model.proposal_6d.detach().cpu().numpy(), g["p6d_f"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(model.proposal_t.detach().cpu().numpy(), g["pt_f"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(model.seg_head.model[2].weight.detach().cpu().numpy()[:, :, 0], g["W2_f"], rtol=0,
                               atol=3e-4)


@pytest.mark.parametrize("robust", [False, True])
def test_fused_step_with_flow_matches_oracle(oracle, dev, robust):
    """Chamfer + flow branch, ragged reference sets, N not a multiple of 64, canonical frame
    in the middle; three iterations against the oracle's iteration."""
    from oracle.step import RelaxOracle
    from reart_amd.relax import RelaxEngine

    rng = np.random.default_rng(8)
    N, P, B, H, cano_idx = 300, 20, 4, 128, 1
    cano = rng.uniform(-0.3, 0.3, (N, 3)).astype(np.float32)
    pcs = (cano[None] + rng.normal(0, 0.02, (B, N, 3))).astype(np.float32)
    W1, b1 = rng.normal(0, 0.6, (H, 3)).astype(np.float32), rng.normal(0, 0.1, H).astype(np.float32)
    W2 = rng.normal(0, 0.2, (P, H)).astype(np.float32)
    p6d = (np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1)) + rng.normal(0, 0.05, (B, P, 6))).astype(np.float32)
    pt = rng.normal(0, 0.01, (B, P, 3)).astype(np.float32)
    lens = [211, 137, 300, 64]
    refs = [rng.uniform(-0.3, 0.3, (m, 3)).astype(np.float32) for m in lens]
    flows = [rng.normal(0, 0.02, (m, 3)).astype(np.float32) for m in lens]
    flows[1][:50] *= 0.001
    orc = RelaxOracle(cano, pcs, W1, b1, W2, p6d, pt, cano_idx, refs, flows, lambda_flow=0.7, robust=robust, n_iter=50)
    model = _make_model(dev, P, B, W1, b1, W2, p6d, pt)
    eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, cano_idx, [t(r, dev) for r in refs], [t(f, dev) for f in flows],
                      n_iter=50, lambda_flow=0.7, use_robust_loss=robust)
    for i in range(3):
        noise = -np.log(rng.exponential(size=(N, P))).astype(np.float32)
        ref = orc.step(noise)
        eng.set_gumbel(t(noise, dev))
        eng.step()
        row = eng.last_losses().cpu().numpy()
        assert abs(row[0] - ref["recon"]) <= 1e-5 * abs(ref["recon"]), (i, row, ref["recon"])
        assert abs(row[1] - ref["flow"]) <= 1e-5 * abs(ref["flow"]) + 1e-9, (i, row, ref["flow"])
        assert abs(row[3] - ref["tau"]) < 1e-6
        np.testing.assert_array_equal(eng.seg_part.cpu().numpy(), ref["seg_part"])
        np.testing.assert_allclose(eng.pc_trans.cpu().numpy(), ref["pc_trans"], rtol=0, atol=5e-7)
        for k, prm in (("p6d", model.proposal_6d), ("pt", model.proposal_t), ("W2", model.seg_head.model[2].weight),
                       ("W1", model.seg_head.model[0].weight), ("b1", model.seg_head.model[0].bias)):
            got = prm.detach().cpu().numpy().reshape(orc.params[k].shape)
            np.testing.assert_allclose(got, orc.params[k], rtol=0, atol=2e-5, err_msg=f"iter {i} param {k}")


def test_graph_replay_equals_eager_and_is_deterministic(dev):
    """In-kernel Philox noise: a captured graph replays the same trajectory as eager launches,
    bit for bit, and the loop never touches the host."""
    from reart_amd.relax import RelaxEngine

    rng = np.random.default_rng(1)
    N, P, B = 1024, 20, 6
    cano = rng.uniform(-0.3, 0.3, (N, 3)).astype(np.float32)
    pcs = (cano[None] + rng.normal(0, 0.02, (B, N, 3))).astype(np.float32)
    refs = [rng.uniform(-0.3, 0.3, (700 + 10 * i, 3)).astype(np.float32) for i in range(B)]
    flows = [rng.normal(0, 0.02, r.shape).astype(np.float32) for r in refs]
    results = []
    for mode in ("eager", "graph", "graph"):
        torch.manual_seed(0)
        from reart_amd.networks.model import BaseModel

        model = BaseModel(num_parts=P, pose_len=B).to(dev)
        eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, 3, [t(r, dev) for r in refs], [t(f, dev) for f in flows],
                          n_iter=100, seed=5)
        done = eng.capture() if mode == "graph" else 0
        eng.step(12 - done)
        it, log = eng.loss_log()
        assert it == 12
        results.append((log.cpu().numpy(), model.proposal_t.detach().cpu().numpy().copy(),
                        model.seg_head.model[2].weight.detach().cpu().numpy().copy()))
    for r in results[1:]:
        for x, y in zip(results[0], r):
            np.testing.assert_array_equal(x, y)
    log = results[0][0]
    assert np.isfinite(log).all()


def test_long_trajectory_stays_finite(dev):
    """3000 iterations with the in-kernel Philox Gumbel noise (6e7 draws): no rare-branch NaN
    (u -> 1.0 gives g = +inf) and the loss log stays finite and decreases."""
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=6, n_parts=4, pts_per_part=256, seed=4, n_ref=600)
    cano, pcs = split_canonical(seq["complete"], 2)
    torch.manual_seed(2)
    model = BaseModel(num_parts=20, pose_len=5).to(dev)
    eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, 2, [t(r, dev) for r in seq["ref_loc"]],
                      [t(f, dev) for f in seq["ref_flow"]], n_iter=3000, ring=4096)
    done = eng.capture()
    eng.step(3000 - done)
    it, log = eng.loss_log()
    log = log.cpu().numpy()
    assert it == 3000 and np.isfinite(log).all()
    assert log[-50:, 2].mean() < 0.6 * log[:50, 2].mean()
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_grid_search_step_is_bit_identical_to_brute_force(dev):
    """The exact grid search over the static targets must not change a single bit of the trajectory."""
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=6, n_parts=4, pts_per_part=300, seed=9, n_ref=777)
    cano, pcs = split_canonical(seq["complete"], 3)
    res = []
    for use_grid in (False, True):
        torch.manual_seed(2)
        model = BaseModel(num_parts=20, pose_len=5).to(dev)
        eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, 3, [t(r, dev) for r in seq["ref_loc"]],
                          [t(f, dev) for f in seq["ref_flow"]], n_iter=200, seed=11, use_grid=use_grid)
        eng.step(25)
        it, log = eng.loss_log()
        res.append((log.cpu().numpy(), eng.pc_trans.cpu().numpy().copy(),
                    [p.detach().cpu().numpy().copy() for p in model.parameters()]))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        np.testing.assert_array_equal(a, b)


def test_search_variants_give_identical_trajectories(dev):
    """The exact searches are interchangeable IN SITU: 40 iterations of the same instance with the
    box-pruned warm-started search (default: sparse scans, three waves per search workgroup, XCD chunks re-sorted by
    measured work), the same with dense scans only, other sparse thresholds, 1 / 2 / 4 waves per workgroup, the static
    launch order, the profiling stamps switched on, the other workgroup geometries of the forward and the backward, and
    cold brute force must leave bit-identical parameters and loss logs."""
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=6, n_parts=4, pts_per_part=300, seed=3, n_ref=700, with_flow=True)
    cano, pcs = split_canonical(seq["complete"], 2)
    runs = {}
    variants = (("pruned", {}), ("cloud1", {"tune_cloud": 1}), ("cloud2_dense", {"tune_cloud": 2, "tune_sparse": -1}), ("cloud8", {"tune_cloud": 8}),
                ("global_targets", {"tune_cloud": 0}), ("global_dense", {"tune_cloud": 0, "tune_sparse": -1}), ("dense", {"tune_sparse": -1}), ("sparse8", {"tune_sparse": 8}),
                ("sparse3", {"tune_sparse": 3}), ("queue64", {"tune_sparse": 64}), ("split1", {"tune_slices": 1, "tune_cloud": 0}), ("split2_flow4", {"tune_slices": 2, "tune_slices_flow": 4, "tune_cloud": 0}),
                ("split4_dense", {"tune_slices": 4, "tune_sparse": -1, "tune_cloud": 0}), ("static_order", {"tune_reorder": -1, "tune_cloud": 0}),
                ("xcd_chunks", {"tune_xcd": 1, "tune_cloud": 0}), ("profiled", {"profile": 1}), ("fwd64", {"tune_fwd_pts": 64}),
                ("brute", {"search_mode": 1}), ("bwd64", {"tune_bwd_pts": 64}), ("bwd16", {"tune_bwd_pts": 16}))
    for name, tuning in variants:
        torch.manual_seed(0)
        model = BaseModel(num_parts=12, pose_len=5).to(dev)
        eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, 2, [t(r, dev) for r in seq["ref_loc"]],
                          [t(f, dev) for f in seq["ref_flow"]], n_iter=200, tuning=tuning)
        eng.step(40)
        it, log = eng.loss_log()
        runs[name] = (log.cpu().numpy(), model.proposal_6d.detach().cpu().numpy().copy(),
                      model.seg_head.model[2].weight.detach().cpu().numpy().copy(), eng.seg_part.cpu().numpy())
        if name == "profiled":
            prof = eng.search_profile(reset=True)
            brute_pairs = 5 * (2 * 1200 * 1200 + 1200 * 700)          # what a brute-force search evaluates per launch
            assert prof["launches"] == 40 and 0 < prof["seconds"] < 40 * 5e-3
            assert 40 * 3 * 5 * 1200 < prof["pairs"] < 2 * 40 * brute_pairs  # more than the seeds; bounded by (padded) brute force
            assert eng.search_profile()["launches"] == 0
    ref = runs["brute"]
    assert np.isfinite(ref[0]).all()
    for name, _ in variants:
        for a, b in zip(runs[name], ref):
            if name.startswith("bwd"):   # other partial-sum chunks in the backward: same sums in another order
                np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6, err_msg=name)
            else:
                np.testing.assert_array_equal(a, b, err_msg=name)


def test_full_size_pruned_equals_brute_force(dev, monkeypatch):
    """BASELINE configuration (T = 20 x N = 4096, P = 20, Chamfer + flow): 12 iterations with the pruned,
    warm-started searches leave exactly the parameters and losses of cold brute-force searches, and the
    engine's per-point output is a permutation-free view (results come back in the caller's order)."""
    import bench

    runs = {}
    for name, env in (("pruned", {}), ("brute", {"REART_SEARCH": "brute"})):
        monkeypatch.delenv("REART_SEARCH", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng, seq, model = bench.build_instance(dev, 20, 4096, 10, seed=2)
        eng.step(12)
        it, log = eng.loss_log()
        runs[name] = (log.cpu().numpy(), model.proposal_6d.detach().cpu().numpy().copy(),
                      model.proposal_t.detach().cpu().numpy().copy(), eng.pc_trans.cpu().numpy(), eng.seg_part.cpu().numpy())
    assert np.isfinite(runs["brute"][0]).all()
    for a, b in zip(runs["pruned"], runs["brute"]):
        np.testing.assert_array_equal(a, b)
    # the losses go down over the first iterations on this well-posed synthetic instance
    assert runs["pruned"][0][-1, 2] < runs["pruned"][0][0, 2]


def test_assignment_loss_step_matches_oracle(oracle, dev):
    """Assignment-loss phase (run_robot.py:164-187, i >= assign_iter): fixed source / target pairs replace the
    Chamfer loss, the flow loss stays.  Two Chamfer iterations, then three assignment iterations with the pairs
    refreshed in between, against the oracle's iteration; peek_forward() returns the coming iteration's clouds."""
    from oracle.step import RelaxOracle
    from reart_amd.relax import RelaxEngine

    rng = np.random.default_rng(21)
    N, P, B, H, cano_idx = 260, 12, 3, 128, 2
    cano = rng.uniform(-0.3, 0.3, (N, 3)).astype(np.float32)
    pcs = (cano[None] + rng.normal(0, 0.02, (B, N, 3))).astype(np.float32)
    W1, b1 = rng.normal(0, 0.6, (H, 3)).astype(np.float32), rng.normal(0, 0.1, H).astype(np.float32)
    W2 = rng.normal(0, 0.2, (P, H)).astype(np.float32)
    p6d = (np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1)) + rng.normal(0, 0.05, (B, P, 6))).astype(np.float32)
    pt = rng.normal(0, 0.01, (B, P, 3)).astype(np.float32)
    refs = [rng.uniform(-0.3, 0.3, (m, 3)).astype(np.float32) for m in (150, 90, 200)]
    flows = [rng.normal(0, 0.02, r.shape).astype(np.float32) for r in refs]
    lam = 0.3
    orc = RelaxOracle(cano, pcs, W1, b1, W2, p6d, pt, cano_idx, refs, flows, lambda_flow=0.7, n_iter=50)
    model = _make_model(dev, P, B, W1, b1, W2, p6d, pt)
    eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, cano_idx, [t(r, dev) for r in refs], [t(f, dev) for f in flows],
                      n_iter=50, lambda_flow=0.7)
    assign = None
    for i in range(5):
        noise = -np.log(rng.exponential(size=(N, P))).astype(np.float32)
        eng.set_gumbel(t(noise, dev))
        if i >= 2:
            if i != 3:   # refresh the pairs at iterations 2 and 4
                src = rng.permutation(N)[:64]
                tgt = np.stack([rng.permutation(N)[:64] for _ in range(B)])
                assign = (src, tgt, lam)
                eng.peek_forward()
                peek = eng.pc_trans.cpu().numpy().copy()
                eng.set_assignment(torch.from_numpy(src), torch.from_numpy(tgt), lam)
        ref = orc.step(noise, assign=assign)
        eng.step()
        if i >= 2 and i != 3:
            np.testing.assert_array_equal(peek, eng.pc_trans.cpu().numpy())     # the step redoes the same forward
        row = eng.last_losses().cpu().numpy()
        assert abs(row[0] - ref["recon"]) <= 1e-5 * abs(ref["recon"]), (i, row, ref["recon"])
        assert abs(row[1] - ref["flow"]) <= 1e-5 * abs(ref["flow"]) + 1e-9, (i, row, ref["flow"])
        np.testing.assert_allclose(eng.pc_trans.cpu().numpy(), ref["pc_trans"], rtol=0, atol=5e-7)
        for k, prm in (("p6d", model.proposal_6d), ("pt", model.proposal_t), ("W2", model.seg_head.model[2].weight),
                       ("W1", model.seg_head.model[0].weight), ("b1", model.seg_head.model[0].bias)):
            got = prm.detach().cpu().numpy().reshape(orc.params[k].shape)
            np.testing.assert_allclose(got, orc.params[k], rtol=0, atol=2e-5, err_msg=f"iter {i} param {k}")


def test_full_size_sparse_scan_equals_dense_scan_over_a_run(dev, monkeypatch):
    """BASELINE configuration, 400 iterations (the state moves through the high-temperature phase): the
    sparse forms of the box scan (8 and 16 queries side by side) leave exactly the loss log and parameters of
    dense scans only."""
    import bench

    runs = {}
    for name, env in (("sparse", {}), ("dense", {"REART_SPARSE": "0"})):
        monkeypatch.delenv("REART_SPARSE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng, seq, model = bench.build_instance(dev, 20, 4096, 10, seed=2)
        eng.step(400)
        it, log = eng.loss_log()
        runs[name] = (log.cpu().numpy(), model.proposal_6d.detach().cpu().numpy().copy(),
                      model.proposal_t.detach().cpu().numpy().copy(), eng.pc_trans.cpu().numpy(), eng.seg_part.cpu().numpy())
    assert np.isfinite(runs["dense"][0]).all()
    for a, b in zip(runs["sparse"], runs["dense"]):
        np.testing.assert_array_equal(a, b)


def test_weight_decay_matches_oracle(oracle, dev):
    """--weight_decay (torch.optim.Adam's L2 form, run_robot.py:146-148) on the fused engine: three iterations against the
    oracle's Adam with grad += wd * param; and it changes the trajectory."""
    from oracle.step import RelaxOracle
    from reart_amd.relax import RelaxEngine

    rng = np.random.default_rng(5)
    N, P, B, H, cano_idx, wd = 200, 8, 3, 128, 1, 0.05
    cano = rng.uniform(-0.3, 0.3, (N, 3)).astype(np.float32)
    pcs = (cano[None] + rng.normal(0, 0.02, (B, N, 3))).astype(np.float32)
    W1, b1 = rng.normal(0, 0.6, (H, 3)).astype(np.float32), rng.normal(0, 0.1, H).astype(np.float32)
    W2 = rng.normal(0, 0.2, (P, H)).astype(np.float32)
    p6d = (np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1)) + rng.normal(0, 0.05, (B, P, 6))).astype(np.float32)
    pt = rng.normal(0, 0.01, (B, P, 3)).astype(np.float32)
    finals = {}
    for decay in (wd, 0.0):
        orc = RelaxOracle(cano, pcs, W1, b1, W2, p6d, pt, cano_idx, n_iter=50, weight_decay=decay)
        model = _make_model(dev, P, B, W1, b1, W2, p6d, pt)
        eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, cano_idx, n_iter=50, weight_decay=decay)
        nrng = np.random.default_rng(9)
        for i in range(3):
            noise = -np.log(nrng.exponential(size=(N, P))).astype(np.float32)
            orc.step(noise)
            eng.set_gumbel(t(noise, dev))
            eng.step()
        for k, prm in (("p6d", model.proposal_6d), ("pt", model.proposal_t), ("W2", model.seg_head.model[2].weight),
                       ("W1", model.seg_head.model[0].weight), ("b1", model.seg_head.model[0].bias)):
            got = prm.detach().cpu().numpy().reshape(orc.params[k].shape)
            np.testing.assert_allclose(got, orc.params[k], rtol=0, atol=2e-5, err_msg=f"wd {decay} param {k}")
        finals[decay] = model.seg_head.model[2].weight.detach().cpu().numpy().copy()
    assert np.abs(finals[wd] - finals[0.0]).max() > 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 3, 6])
def test_batched_instances_equal_separate_engines(dev, K):
    """reart_relax_step_batch: K instances of one shape (their own canonical frame, clouds, parameters, seeds) stepping in
    shared launches -- eagerly and replayed from one graph -- leave bit for bit what K separate engines leave."""
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxBatch, RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=6, n_parts=4, pts_per_part=300, seed=5, n_ref=700, with_flow=True)

    def build():
        out = []
        for k in range(K):
            ci = k % 6
            cano, pcs = split_canonical(seq["complete"], ci)
            torch.manual_seed(10 + k)
            model = BaseModel(num_parts=12, pose_len=5).to(dev)
            refs = [t(r, dev) for r in seq["ref_loc"]], [t(f, dev) for f in seq["ref_flow"]]
            out.append((RelaxEngine(t(cano, dev), t(pcs, dev), model, ci, refs[0], refs[1], n_iter=200, seed=100 + k), model))
        return out

    def state(eng, model):
        it, log = eng.loss_log()
        return (log.cpu().numpy(), model.proposal_6d.detach().cpu().numpy().copy(), model.proposal_t.detach().cpu().numpy().copy(),
                model.seg_head.model[2].weight.detach().cpu().numpy().copy(), eng.pc_trans.cpu().numpy(), eng.seg_part.cpu().numpy())

    solo = build()
    for eng, _ in solo:
        eng.step(36)
    batched = build()
    batch = RelaxBatch([e for e, _ in batched])
    batch.step(11)                       # eager
    batch.capture(steps_per_graph=4)     # +1 (warm-up)
    batch.step(24)                       # 6 replays
    assert batch.graph_replays == 6 and batch.eager_steps == 11
    torch.cuda.synchronize()
    for (e0, m0), (e1, m1) in zip(solo, batched):
        a, b = state(e0, m0), state(e1, m1)
        assert np.isfinite(a[0]).all() and a[0].shape == b[0].shape
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    if K > 1:    # the instances really are different problems
        assert not np.array_equal(state(*solo[0])[1], state(*solo[1])[1])
    with pytest.raises(ValueError):
        RelaxBatch([])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["chamfer_only", "assign_flow", "assign_only"])
def test_batched_instances_in_the_unmerged_modes(dev, mode):
    """reart_relax_step_batch beyond the merged Chamfer + flow iteration: Chamfer only, and the assignment loss (pairs set
    with set_assignment, refreshed once on the way) with and without the flow loss -- the second phase of the README recipe
    (run_robot.py:164-192).  Three instances in shared launches, eager and replayed, bit for bit what separate engines leave."""
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxBatch, RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    K, flow = 3, mode == "assign_flow"
    seq = make_sequence(T=6, n_parts=4, pts_per_part=300, seed=5, n_ref=700, with_flow=True)
    N = seq["complete"].shape[1]
    rng = np.random.default_rng(7)
    pairs = [[(rng.permutation(N)[:256], np.stack([rng.permutation(N)[:256] for _ in range(5)])) for _ in range(2)] for _ in range(K)]

    def build():
        out = []
        for k in range(K):
            cano, pcs = split_canonical(seq["complete"], k)
            torch.manual_seed(10 + k)
            model = BaseModel(num_parts=12, pose_len=5).to(dev)
            refs = ([t(r, dev) for r in seq["ref_loc"]], [t(f, dev) for f in seq["ref_flow"]]) if flow else (None, None)
            out.append((RelaxEngine(t(cano, dev), t(pcs, dev), model, k, refs[0], refs[1], n_iter=200, seed=100 + k), model))
        return out

    def assign(engines, r):
        if mode != "chamfer_only":
            for k, (e, _) in enumerate(engines):
                e.set_assignment(torch.from_numpy(pairs[k][r][0]), torch.from_numpy(pairs[k][r][1]), 0.3)

    def state(eng, model):
        it, log = eng.loss_log()
        return (log.cpu().numpy(), model.proposal_6d.detach().cpu().numpy().copy(), model.proposal_t.detach().cpu().numpy().copy(),
                model.seg_head.model[2].weight.detach().cpu().numpy().copy(), eng.pc_trans.cpu().numpy(), eng.seg_part.cpu().numpy())

    solo = build()
    assign(solo, 0)
    for e, _ in solo:
        e.step(8)
    assign(solo, 1)
    for e, _ in solo:
        e.step(9)
    batched = build()
    assign(batched, 0)
    batch = RelaxBatch([e for e, _ in batched])
    batch.step(8)                        # eager
    assign(batched, 1)                   # the pairs live in each engine's buffer: a refresh needs no new batch
    batch.capture(steps_per_graph=4)     # +1 (warm-up)
    batch.step(8)                        # 2 replays
    torch.cuda.synchronize()
    for (e0, m0), (e1, m1) in zip(solo, batched):
        a, b = state(e0, m0), state(e1, m1)
        assert np.isfinite(a[0]).all()
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
