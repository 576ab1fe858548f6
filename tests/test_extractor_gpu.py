"""GPU parity of the PointNet++ correspondence extractor (MFMA GEMM stacks, gather fusion, max-pool,
3-NN interpolation) against the reference's PointNet2Msg2 run on CPU with the same seeded weights (G9)."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_rules")]   # goldens follow the CPU-fallback rules
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_mlp_layer_vs_torch(dev):
    """The MFMA layer alone: odd sizes, bias, relu, pooling, column-offset output, gathered input."""
    from reart_amd.networks.feature_extractor import mlp_layer

    rng = np.random.default_rng(0)
    for rows, cin, cout, pool in ((300, 6, 32, 0), (512, 323, 196, 64), (128, 515, 256, 128), (1000, 134, 128, 0),
                                  (96, 17, 20, 32)):
        X = rng.normal(size=(rows, cin)).astype(np.float32)
        # asymmetric weights: a transposed fragment layout cannot pass
        W = (rng.normal(size=(cin, cout)) + np.arange(cout)[None, :] * 0.01).astype(np.float32)
        b = rng.normal(size=cout).astype(np.float32)
        ref = np.maximum(X.astype(np.float64) @ W.astype(np.float64) + b, 0)
        if pool:
            ref = ref.reshape(rows // pool, pool, cout).max(axis=1)
        out = torch.full((ref.shape[0], cout + 5), -7.0, device=dev)
        mlp_layer(t(X, dev), t(W, dev), t(b, dev), relu=True, pool_k=pool, out=out, out_col=3)
        got = out.cpu().numpy()
        np.testing.assert_allclose(got[:, 3:3 + cout], ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
        assert (got[:, :3] == -7).all() and (got[:, 3 + cout:] == -7).all()
    # gathered input == explicit grouping
    B, Npts, S, K, D = 2, 50, 8, 32, 5
    F = rng.normal(size=(B * Npts, D)).astype(np.float32)
    Q = rng.normal(size=(B * Npts, 3)).astype(np.float32)
    C = rng.normal(size=(B * S, 3)).astype(np.float32)
    idx = rng.integers(0, Npts, (B, S, K))
    W = rng.normal(size=(D + 3, 24)).astype(np.float32)
    b = rng.normal(size=24).astype(np.float32)
    rows_f = F.reshape(B, Npts, D)[np.arange(B)[:, None, None], idx]
    rows_q = Q.reshape(B, Npts, 3)[np.arange(B)[:, None, None], idx] - C.reshape(B, S, 1, 3)
    for xyz_first in (0, 1):
        Xg = np.concatenate([rows_q + (C.reshape(B, S, 1, 3) if xyz_first else 0), rows_f] if xyz_first
                            else [rows_f, rows_q], axis=-1).reshape(-1, D + 3)
        ref = np.maximum(Xg.astype(np.float64) @ W + b, 0).reshape(B * S, K, 24).max(axis=1)
        got = mlp_layer(None, t(W, dev), t(b, dev), pool_k=K,
                        gather=dict(idx=t(idx, dev), F=t(F, dev), Q=t(Q, dev), C=None if xyz_first else t(C, dev),
                                    Npts=Npts, xyz_first=xyz_first))
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


def test_extractor_matches_reference(dev):
    from reart_amd.networks.feature_extractor import PointNet2Msg2
    from reart_amd.synthetic import extractor_state

    g = np.load(os.path.join(G, "extractor.npz"))
    model = PointNet2Msg2(out_dim=64)
    model.load_state_dict(extractor_state(model), strict=True)
    model = model.to(dev).eval()
    xyz = t(g["xyz"], dev)
    pts = xyz.permute(0, 2, 1).contiguous()
    # stage-wise: sa1 / sa2 outputs (channel-first in the reference)
    l1_xyz, l1 = model.sa1.run(pts, pts, start=t(g["start1"], dev))
    np.testing.assert_array_equal(l1_xyz.permute(0, 2, 1).cpu().numpy(), g["l1_xyz"])
    np.testing.assert_allclose(l1.permute(0, 2, 1).cpu().numpy(), g["l1_points"], rtol=1e-4, atol=1e-4)
    l2_xyz, l2 = model.sa2.run(l1_xyz, l1, start=t(g["start2"], dev))
    np.testing.assert_array_equal(l2_xyz.permute(0, 2, 1).cpu().numpy(), g["l2_xyz"])
    np.testing.assert_allclose(l2.permute(0, 2, 1).cpu().numpy(), g["l2_points"], rtol=1e-4, atol=1e-4)
    feat = model(xyz, fps_start=(t(g["start1"], dev), t(g["start2"], dev)))
    assert tuple(feat.shape) == (2, 64, 1024)
    ref = g["feat"]
    err = np.abs(feat.cpu().numpy() - ref)
    # ball-query membership and the 3-NN weights are the reference's bit for bit (matmul-expanded distances with
    # torch's CPU rounding); what is left is fp32 accumulation order (MFMA tiles vs the reference's sgemm)
    print(f"\n[a9 @ N=1024] descriptor error: max {err.max() / np.abs(ref).max():.3e} of the scale, mean {err.mean() / np.abs(ref).mean():.3e} of the mean magnitude")
    assert err.max() <= 2e-5 * np.abs(ref).max(), (err.max(), np.abs(ref).max())
    assert err.mean() <= 5e-6 * np.abs(ref).mean(), (err.mean(), np.abs(ref).mean())


def test_fused_sa1_chain_equals_the_three_launch_path(dev):
    """reart_mlp_chain3 (three layers + pooling in one launch, activations in LDS) against three reart_mlp_layer launches:
    every accumulator sees the same k order through the same MFMA instruction, so the outputs are equal BIT FOR BIT --
    for each of the extractor's three sa1 scales, full and ragged tile counts."""
    from reart_amd.networks import feature_extractor as fe

    rng = np.random.default_rng(5)
    for (C1, C2, C3, K), (B, S, Npts) in (((32, 32, 64, 32), (2, 37, 300)), ((64, 64, 128, 64), (3, 21, 257)),
                                         ((64, 96, 128, 128), (2, 9, 400)), ((64, 96, 128, 128), (5, 64, 1024))):
        F = rng.normal(size=(B * Npts, 3)).astype(np.float32)
        C = rng.normal(size=(B * S, 3)).astype(np.float32)
        idx = rng.integers(0, Npts, (B, S, K))
        folded = []
        cin = 6
        for cout in (C1, C2, C3):
            folded.append((t((rng.normal(size=(cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32), dev),
                           t(rng.normal(0, 0.1, cout).astype(np.float32), dev)))
            cin = cout
        g = dict(idx=t(idx, dev), F=t(F, dev), Q=t(F, dev), C=t(C, dev), Npts=Npts, xyz_first=0)
        h = fe.mlp_layer(None, *folded[0], gather=g)
        h = fe.mlp_layer(h, *folded[1])
        ref = torch.full((B * S, C3 + 7), -3.0, device=dev)
        fe.mlp_layer(h, *folded[2], pool_k=K, out=ref, out_col=5)
        got = torch.full((B * S, C3 + 7), -3.0, device=dev)
        fe.mlp_chain3(folded, g, got, 5)
        assert torch.equal(got, ref), (C1, C2, C3, K, float((got - ref).abs().max()))
        # against float64 on the host as well
        X = np.concatenate([F.reshape(B, Npts, 3)[np.arange(B)[:, None, None], idx],
                            F.reshape(B, Npts, 3)[np.arange(B)[:, None, None], idx] - C.reshape(B, S, 1, 3)], -1).reshape(-1, 6).astype(np.float64)
        for W, bvec in folded:
            X = np.maximum(X @ W.cpu().numpy().astype(np.float64) + bvec.cpu().numpy(), 0)
        want = X.reshape(B * S, K, C3).max(1)
        np.testing.assert_allclose(got[:, 5:5 + C3].cpu().numpy(), want, rtol=2e-5, atol=2e-5 * np.abs(want).max())


def test_fused_sa2_chain_equals_the_three_launch_path(dev):
    """reart_mlp_chain3_wide (sa2's scales: 323 -> 128 -> 128 | 196 -> 256, weights streamed through LDS) against three
    reart_mlp_layer launches: the same bits; and float64 on the host."""
    from reart_amd.networks import feature_extractor as fe

    rng = np.random.default_rng(6)
    for (C1, C2, C3, K), (B, S, Npts, D) in (((128, 128, 256, 64), (2, 6, 70, 320)), ((128, 196, 256, 128), (3, 5, 200, 320)),
                                            ((128, 196, 256, 128), (2, 128, 512, 320)), ((128, 128, 256, 64), (1, 2, 64, 8))):
        F = rng.normal(size=(B * Npts, D)).astype(np.float32)
        Q = rng.normal(size=(B * Npts, 3)).astype(np.float32)
        C = rng.normal(size=(B * S, 3)).astype(np.float32)
        idx = rng.integers(0, Npts, (B, S, K))
        folded = []
        cin = D + 3
        for cout in (C1, C2, C3):
            folded.append((t((rng.normal(size=(cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32), dev),
                           t(rng.normal(0, 0.1, cout).astype(np.float32), dev)))
            cin = cout
        g = dict(idx=t(idx, dev), F=t(F, dev), Q=t(Q, dev), C=t(C, dev), Npts=Npts, xyz_first=0)
        h = fe.mlp_layer(None, *folded[0], gather=g)
        h = fe.mlp_layer(h, *folded[1])
        ref = torch.full((B * S, C3 + 7), -3.0, device=dev)
        fe.mlp_layer(h, *folded[2], pool_k=K, out=ref, out_col=5)
        got = torch.full((B * S, C3 + 7), -3.0, device=dev)
        fe.mlp_chain3_wide(folded, g, got, 5)
        assert torch.equal(got, ref), (C1, C2, C3, K, float((got - ref).abs().max()))
        Fg = F.reshape(B, Npts, D)[np.arange(B)[:, None, None], idx]
        Qg = Q.reshape(B, Npts, 3)[np.arange(B)[:, None, None], idx] - C.reshape(B, S, 1, 3)
        X = np.concatenate([Fg, Qg], -1).reshape(-1, D + 3).astype(np.float64)
        for W, bvec in folded:
            X = np.maximum(X @ W.cpu().numpy().astype(np.float64) + bvec.cpu().numpy(), 0)
        want = X.reshape(B * S, K, C3).max(1)
        np.testing.assert_allclose(got[:, 5:5 + C3].cpu().numpy(), want, rtol=5e-5, atol=5e-5 * np.abs(want).max())


def test_extractor_same_bits_with_and_without_the_fused_chain(dev):
    from reart_amd.networks import feature_extractor as fe
    from reart_amd.synthetic import extractor_state

    g = np.load(os.path.join(G, "extractor.npz"))
    model = fe.PointNet2Msg2(out_dim=64)
    model.load_state_dict(extractor_state(model), strict=True)
    model = model.to(dev).eval()
    xyz = t(g["xyz"], dev)
    starts = (t(g["start1"], dev), t(g["start2"], dev))
    fused = model(xyz, fps_start=starts)
    fe.FUSE_CHAIN = False
    try:
        plain = model(xyz, fps_start=starts)
    finally:
        fe.FUSE_CHAIN = True
    assert torch.equal(fused, plain)
    # and with the second level's sampling on the main stream instead of the side stream
    fe.OVERLAP_SAMPLING = False
    try:
        serial = model(xyz, fps_start=starts)
    finally:
        fe.OVERLAP_SAMPLING = True
    assert torch.equal(fused, serial)
    for _ in range(3):                       # repeated forwards: the side stream's tensors are recycled safely
        assert torch.equal(model(xyz, fps_start=starts), fused)


def test_extractor_vs_oracle_other_size_cuda_rules(oracle, dev):
    """Beyond the fixtures: a 2048-point cloud pair, other seeds, the CUDA sampling rules (FPS from index 0, d2 < r2) -- the HIP
    extractor against the oracle's restatement of the whole forward (oracle/extractor.py, itself pinned to the reference by
    extractor.npz)."""
    from oracle import extractor as ox
    from reart_amd.networks.feature_extractor import PointNet2Msg2
    from reart_amd.synthetic import extractor_state, make_sequence

    seq = make_sequence(T=2, n_parts=4, pts_per_part=512, seed=9, with_flow=False)
    pts = torch.from_numpy(seq["complete"]).float()
    pts = pts - pts.mean(dim=1, keepdim=True)
    pts = pts / pts.norm(dim=-1).max()
    xyz = pts.permute(0, 2, 1).contiguous()
    model = PointNet2Msg2(out_dim=64)
    sd = extractor_state(model, seed=23)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    got = model(xyz.to(dev), cuda_mode=True).cpu().numpy()
    ref = ox.forward({k: v.numpy() for k, v in sd.items()}, xyz.numpy(), cuda_mode=True)
    err = np.abs(got - ref)
    # a ball-query row on the radius boundary may group differently under fp32 rounding of d2 (CUDA rule: coordinate
    # differences, same expression on both sides here), so the bound is the fixtures' one
    assert err.max() <= 1e-5 * np.abs(ref).max(), (err.max(), np.abs(ref).max())
    assert err.mean() <= 5e-6 * np.abs(ref).mean()
