"""Round-2 parity tests on the GPU:
  * a12 3-NN interpolation alone, against the reference's own module (golden) and the oracle, bit-exact
  * a14 knn_query: k = 1 / 3 / 5 label modes and the 2-D mean branch, against the reference (golden)
  * a9  the extractor at the BASELINE cloud size N = 4096, CPU-fallback and CUDA sampling rules
  * a16 the fused step at T = 20 x N = 4096 against the oracle's step, directly
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _three_nn(xyz1, xyz2):
    from reart_amd import _lib

    B, N, _ = xyz1.shape
    d = torch.empty((B, N, 3), dtype=torch.float32, device=xyz1.device)
    i = torch.empty((B, N, 3), dtype=torch.int64, device=xyz1.device)
    rc = _lib.lib().reart_three_nn(_lib.ptr(xyz1), _lib.ptr(xyz2), B, N, xyz2.shape[1], _lib.ptr(d), _lib.ptr(i),
                                   _lib.stream())
    _lib.check(rc, "reart_three_nn")
    return d, i


@pytest.mark.parametrize("tag", ["a", "b"])
def test_three_interpolate_reference_golden(dev, tag):
    """(N,S,D) = (512,128,512) and (4096,512,256): indices, distances and interpolated features BIT-equal to
    the reference's PointNetFeaturePropagation (empty mlp) on its CPU path, negative 'zero' distances included."""
    from reart_amd.networks.feature_extractor import three_interpolate

    g = np.load(os.path.join(G, "three_interp.npz"))
    x1, x2 = t(g[tag + "_xyz1"], dev), t(g[tag + "_xyz2"], dev)
    p2 = t(g[tag + "_points2_f16"].astype(np.float32), dev)
    d, i = _three_nn(x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), g[tag + "_i3"])
    np.testing.assert_array_equal(d.cpu().numpy(), g[tag + "_d3"])
    assert (g[tag + "_d3"] < 0).any()          # the case that separates the two distance forms is in the fixture
    B, N, D = x1.shape[0], x1.shape[1], p2.shape[2]
    out = torch.full((B * N, D + 7), -3.0, device=dev)
    three_interpolate(x1, x2, p2, out, 4)
    got = out.reshape(B, N, D + 7).cpu().numpy()
    assert (got[..., :4] == -3).all() and (got[..., 4 + D:] == -3).all()
    np.testing.assert_array_equal(got[:, ::int(g[tag + "_stride"]), 4:4 + D], g[tag + "_out"])


def test_three_interpolate_vs_oracle_and_distance_form(oracle, dev):
    """Ragged sizes against the oracle (bit-exact), and a host-side measurement of what the distance form is worth:
    direct-difference distances move the interpolation by < 1e-3 absolute (coincident / near-coincident points)."""
    from reart_amd.networks.feature_extractor import three_interpolate

    rng = np.random.default_rng(4)
    for B, N, S, D in ((1, 5, 3, 1), (3, 777, 130, 33), (2, 2500, 1025, 8)):
        x1 = rng.uniform(-0.5, 0.5, (B, N, 3)).astype(np.float32)
        x2 = np.ascontiguousarray(x1[:, rng.permutation(N)[:S]]) if S <= N else rng.uniform(-0.5, 0.5, (B, S, 3)).astype(np.float32)
        if S > 10:
            x2[:, 7] = x2[:, 3]                 # duplicated coarse points: (d, index) order
        p2 = rng.normal(size=(B, S, D)).astype(np.float32)
        ref = oracle.three_interpolate(x1, x2, p2)
        rd, ri = oracle.three_nn_expanded(x1, x2)
        d, i = _three_nn(t(x1, dev), t(x2, dev))
        np.testing.assert_array_equal(i.cpu().numpy(), ri)
        np.testing.assert_array_equal(d.cpu().numpy(), rd)
        out = torch.empty((B * N, D), device=dev)
        three_interpolate(t(x1, dev), t(x2, dev), t(p2, dev), out, 0)
        np.testing.assert_array_equal(out.reshape(B, N, D).cpu().numpy(), ref)
    # distance-form experiment on the golden's own inputs
    g = np.load(os.path.join(G, "three_interp.npz"))
    x1, x2 = g["a_xyz1"], g["a_xyz2"]
    p2 = g["a_points2_f16"].astype(np.float32)
    d_direct, i_direct = oracle.knn_points(x1, x2, K=3)
    w = 1.0 / (d_direct + np.float32(1e-8))
    w = w / w.sum(-1, keepdims=True)
    direct = (p2[np.arange(2)[:, None, None], i_direct] * w[..., None]).sum(2)[:, ::int(g["a_stride"])]
    err = np.abs(direct - g["a_out"]).max(-1)                       # per query
    coincident = (d_direct[:, ::int(g["a_stride"]), 0] == 0)
    assert coincident.sum() >= 100
    # the two forms are NOT interchangeable bit-wise, but the gap is 1e-5 .. 1e-3 absolute on O(1) features (measured
    # 8e-5): the distance form was not what round 1's 2e-3 extractor tolerance absorbed -- ball-query rows on the
    # radius boundary were (the CPU-fallback ball query now evaluates the reference's expanded distance too)
    assert 1e-5 < err.max() < 1e-3


def test_knn_query_reference_golden(dev):
    from reart_amd.knn_cuda import KNN
    from reart_amd.utils.model_utils import knn_query

    g = np.load(os.path.join(G, "knn_query.npz"))
    src, query = t(g["src"], dev), t(g["query"], dev)
    labels, feats = t(g["labels"], dev), t(g["feats"], dev)
    for k in (1, 3, 5):
        got = knn_query(query, src, labels, KNN(k=k, transpose_mode=True))
        assert got.dtype == torch.int64
        np.testing.assert_array_equal(got.cpu().numpy(), g[f"labels_k{k}"])
    got = knn_query(query[:700].contiguous(), src, labels, KNN(k=3, transpose_mode=True))
    np.testing.assert_array_equal(got.cpu().numpy(), g["labels_k3_short"])
    got = knn_query(query, src, feats, KNN(k=3, transpose_mode=True))
    np.testing.assert_allclose(got.cpu().numpy(), g["feats_k3"], rtol=0, atol=1e-6)
    with pytest.raises(RuntimeError):          # the reference's reshape fails the same way (n_query != n_src)
        knn_query(query[:700].contiguous(), src, feats, KNN(k=3, transpose_mode=True))


def test_row_mode_matches_torch_cpu(dev):
    from reart_amd.utils.model_utils import _row_mode

    rng = np.random.default_rng(0)
    for k in (1, 2, 3, 5, 20):
        v = rng.integers(0, 4, (500, k))
        np.testing.assert_array_equal(_row_mode(t(v, dev)).cpu().numpy(), torch.mode(torch.from_numpy(v), dim=1)[0].numpy())


EXTRACTOR_TOL = 1e-5      # of the descriptors' scale (north_star: fp32 within 1e-4 relative); measured on MI355X: 1.6e-6 (printed below)


def _extractor(dev):
    from reart_amd.networks.feature_extractor import PointNet2Msg2
    from reart_amd.synthetic import extractor_state

    model = PointNet2Msg2(out_dim=64)
    model.load_state_dict(extractor_state(model), strict=True)
    return model.to(dev).eval()


def test_extractor_4096_reference_golden(dev):
    """The BASELINE extractor shape (flow_utils.py:123-124: clouds of 4096 points) against the reference's own
    PointNet2Msg2: sampled coordinates bit-equal, descriptors within 1e-5 of their scale (measured 1.6e-6) (fp32 MFMA accumulation
    order vs the reference's sgemm; the 3-NN weights are the reference's bit for bit)."""
    g = np.load(os.path.join(G, "extractor_4096.npz"))
    model = _extractor(dev)
    xyz = t(g["xyz"], dev)
    pts = xyz.permute(0, 2, 1).contiguous()
    l1_xyz, l1 = model.sa1.run(pts, pts, start=t(g["start1"], dev), cuda_mode=False)
    np.testing.assert_array_equal(l1_xyz.permute(0, 2, 1).cpu().numpy(), g["l1_xyz"])
    l2_xyz, l2 = model.sa2.run(l1_xyz, l1, start=t(g["start2"], dev), cuda_mode=False)
    np.testing.assert_array_equal(l2_xyz.permute(0, 2, 1).cpu().numpy(), g["l2_xyz"])
    np.testing.assert_allclose(l2.permute(0, 2, 1).cpu().numpy(), g["l2_points"], rtol=1e-4, atol=1e-4)
    feat = model(xyz, fps_start=(t(g["start1"], dev), t(g["start2"], dev)), cuda_mode=False).cpu().numpy()
    ref = g["feat"]
    assert feat.shape == (1, 64, 4096)
    err = np.abs(feat - ref)
    l2e = np.abs(l2.permute(0, 2, 1).cpu().numpy() - g["l2_points"])
    print(f"\n[a9 @ N=4096, CPU rules] descriptor error: max {err.max():.3e} = {err.max() / np.abs(ref).max():.3e} of the scale "
          f"({np.abs(ref).max():.3f}), mean {err.mean():.3e} = {err.mean() / np.abs(ref).mean():.3e} of the mean magnitude; "
          f"sa2 output: max {l2e.max():.3e} of scale {np.abs(g['l2_points']).max():.3f}")
    assert err.max() <= EXTRACTOR_TOL * np.abs(ref).max(), (err.max(), np.abs(ref).max())
    assert err.mean() <= 5e-6 * np.abs(ref).mean()
    # CUDA sampling rules (what the reference computes on a GPU): start 0, d2 < r2, padded with the first hit
    feat_c = model(xyz).cpu().numpy()            # the package default: pointnet2_utils.CUDA = True
    l1c, _ = model.sa1.run(pts, pts)
    np.testing.assert_array_equal(l1c.permute(0, 2, 1).cpu().numpy(), g["cuda_l1_xyz"])
    ref = g["cuda_feat"]
    err = np.abs(feat_c - ref)
    print(f"[a9 @ N=4096, CUDA rules] descriptor error: max {err.max():.3e} = {err.max() / np.abs(ref).max():.3e} of the scale, "
          f"mean {err.mean():.3e} = {err.mean() / np.abs(ref).mean():.3e} of the mean magnitude")
    assert err.max() <= EXTRACTOR_TOL * np.abs(ref).max(), (err.max(), np.abs(ref).max())
    assert not np.array_equal(g["cuda_feat"], g["feat"])


def test_fused_step_full_size_vs_oracle(oracle, dev):
    """BASELINE.json configs[1] AT ITS OWN SIZE (T = 20 x N = 4096, Chamfer + flow, 3000 references per pair):
    two iterations of the fused engine (pruned search, k-d storage order, five launches) against the oracle's
    iteration on the same injected Gumbel noise -- losses 1e-5, transformed clouds 5e-7, part labels equal,
    parameters after Adam 2e-5 -- and the brute-force Chamfer operator on the same transformed clouds against the
    oracle's k-NN: indices and distances bit-equal at 19 x 4096 x 4096."""
    from oracle.step import RelaxOracle
    from reart_amd.chamferdist_C import chamfer_bidir
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    T, N, P, cano_idx = 20, 4096, 20, 10
    seq = make_sequence(T=T, n_parts=8, pts_per_part=N // 8, seed=2, n_ref=3000, with_flow=True)
    cano, pcs = split_canonical(seq["complete"], cano_idx)
    torch.manual_seed(2)
    model = BaseModel(num_parts=P, pose_len=T - 1).to(dev)
    # parts start from DIFFERENT poses: with the reference's identical initial poses the loss does not depend on the
    # segmentation, dL/d(seg head) is pure rounding noise, and Adam's first step (g / (|g| + eps)) amplifies that
    # noise to +-lr in any implementation -- nothing a tolerance could pin
    prng = np.random.default_rng(7)
    with torch.no_grad():
        model.proposal_6d.add_(t(prng.normal(0, 0.05, tuple(model.proposal_6d.shape)).astype(np.float32), dev))
        model.proposal_t.add_(t(prng.normal(0, 0.01, tuple(model.proposal_t.shape)).astype(np.float32), dev))
    c1, c2 = model.seg_head.model[0], model.seg_head.model[2]
    W1, b1, W2 = (c1.weight.detach().cpu().numpy()[:, :, 0].copy(), c1.bias.detach().cpu().numpy().copy(),
                  c2.weight.detach().cpu().numpy()[:, :, 0].copy())
    orc = RelaxOracle(cano, pcs, W1, b1, W2, model.proposal_6d.detach().cpu().numpy(),
                      model.proposal_t.detach().cpu().numpy(), cano_idx, seq["ref_loc"], seq["ref_flow"], n_iter=15000)
    eng = RelaxEngine(t(cano, dev), t(pcs, dev), model, cano_idx, [t(r, dev) for r in seq["ref_loc"]],
                      [t(f, dev) for f in seq["ref_flow"]], n_iter=15000)
    rng = np.random.default_rng(0)
    for i in range(2):
        noise = -np.log(rng.exponential(size=(N, P))).astype(np.float32)
        ref = orc.step(noise)
        eng.set_gumbel(t(noise, dev))
        eng.step()
        row = eng.last_losses().cpu().numpy()
        assert abs(row[0] - ref["recon"]) <= 1e-5 * abs(ref["recon"]), (i, row, ref["recon"])
        assert abs(row[1] - ref["flow"]) <= 1e-5 * abs(ref["flow"]) + 1e-9, (i, row, ref["flow"])
        assert abs(row[3] - ref["tau"]) < 1e-6
        np.testing.assert_array_equal(eng.seg_part.cpu().numpy(), ref["seg_part"])
        pc_trans = eng.pc_trans
        np.testing.assert_allclose(pc_trans.cpu().numpy(), ref["pc_trans"], rtol=0, atol=5e-7)
        for k, prm in (("p6d", model.proposal_6d), ("pt", model.proposal_t), ("W2", model.seg_head.model[2].weight),
                       ("W1", model.seg_head.model[0].weight), ("b1", model.seg_head.model[0].bias)):
            got = prm.detach().cpu().numpy().reshape(orc.params[k].shape)
            np.testing.assert_allclose(got, orc.params[k], rtol=0, atol=2e-5, err_msg=f"iter {i} param {k}")
    # the stand-alone Chamfer operator at full size on the engine's own output
    x = pc_trans.contiguous()
    d_xy, i_xy, d_yx, i_yx = chamfer_bidir(x, t(pcs, dev))
    xn = x.cpu().numpy()
    rd1, ri1 = oracle.knn_points(xn, pcs)
    rd2, ri2 = oracle.knn_points(pcs, xn)
    np.testing.assert_array_equal(i_xy.cpu().numpy(), ri1[..., 0])
    np.testing.assert_array_equal(i_yx.cpu().numpy(), ri2[..., 0])
    np.testing.assert_array_equal(d_xy.cpu().numpy(), rd1[..., 0])
    np.testing.assert_array_equal(d_yx.cpu().numpy(), rd2[..., 0])
