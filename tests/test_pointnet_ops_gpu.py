"""GPU parity of FPS and ball query: bit-exact indices vs the oracle and vs the golden vectors
produced by the reference's own CPU fallbacks (tests/golden/pointnet_ops.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_rules")]   # goldens follow the CPU-fallback rules
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_fps_and_ball_query_reference_golden(dev):
    from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points, query_ball_point

    g = np.load(os.path.join(G, "pointnet_ops.npz"))
    xyz = t(g["xyz"], dev)
    f1 = farthest_point_sample(xyz, 512, start=t(g["start1"], dev))
    np.testing.assert_array_equal(f1.cpu().numpy(), g["fps1"])
    new = index_points(xyz, f1)
    f2 = farthest_point_sample(new, 128, start=t(g["start2"], dev))
    np.testing.assert_array_equal(f2.cpu().numpy(), g["fps2"])
    new2 = index_points(new, f2)
    for r, K, src, ctr, tag in ((0.05, 32, xyz, new, "a"), (0.1, 64, xyz, new, "b"), (0.2, 128, xyz, new, "c"),
                                (0.2, 64, new, new2, "d"), (0.4, 128, new, new2, "e")):
        idx = query_ball_point(r, K, src, ctr)
        assert idx.dtype == torch.int64
        np.testing.assert_array_equal(idx.cpu().numpy(), g["bq_" + tag])


@pytest.mark.parametrize("cuda_mode", [False, True])
@pytest.mark.parametrize("N,M", [(1, 1), (70, 70), (300, 50), (512, 128), (513, 40), (1000, 33), (1025, 30), (2049, 20), (4096, 1024),
                                 (5000, 64), (8192, 16), (12288, 12)])      # every (threads, points per thread) instantiation
def test_fps_vs_oracle(oracle, dev, cuda_mode, N, M):
    from reart_amd.networks.pointnet2_utils import farthest_point_sample

    rng = np.random.default_rng(N + M)
    xyz = rng.uniform(-1, 1, (3, N, 3)).astype(np.float32)
    if N >= 70:
        xyz[:, 5] = xyz[:, 40]  # duplicated points: exercises the arg-max tie rules
        xyz[1, : N // 2] = np.round(xyz[1, : N // 2] * 4) / 4  # lattice: many exact distance ties
    start = rng.integers(0, N, 3).astype(np.int32)
    ref = oracle.fps(xyz, M, start=start, cuda_mode=cuda_mode)
    got = farthest_point_sample(t(xyz, dev), M, start=t(start, dev), cuda_mode=cuda_mode)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("cuda_mode", [False, True])
def test_fps_all_ties(oracle, dev, cuda_mode):
    """Every distance equal (coincident points; a two-value lattice): the whole sequence is decided by the reference's tie
    rule -- lowest thread of the CUDA block then lowest index, or first maximum -- which the kernel carries as a key."""
    from reart_amd.networks.pointnet2_utils import farthest_point_sample

    rng = np.random.default_rng(77)
    for N, M in ((640, 40), (4096, 24)):
        xyz = np.zeros((3, N, 3), np.float32)
        xyz[1] = rng.integers(0, 2, (N, 3)).astype(np.float32)            # the corners of a cube, many copies of each
        xyz[2, ::3] = 1.0
        start = np.array([0, N - 1, N // 2], np.int32)
        ref = oracle.fps(xyz, M, start=start, cuda_mode=cuda_mode)
        got = farthest_point_sample(t(xyz, dev), M, start=t(start, dev), cuda_mode=cuda_mode)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("cuda_mode", [False, True])
def test_ball_query_vs_oracle(oracle, dev, cuda_mode):
    from reart_amd.networks.pointnet2_utils import query_ball_point

    rng = np.random.default_rng(3)
    xyz = rng.uniform(-1, 1, (2, 1500, 3)).astype(np.float32)
    ctr = np.concatenate([xyz[:, :100], rng.uniform(-3, 3, (2, 30, 3)).astype(np.float32)], axis=1)  # some empty balls
    for r, K in ((0.05, 16), (0.3, 32), (0.3, 200), (5.0, 64)):
        ref = oracle.ball_query(r, K, xyz, ctr, cuda_mode=cuda_mode)
        got = query_ball_point(r, K, t(xyz, dev), t(ctr, dev), cuda_mode=cuda_mode)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_pointnet2_cuda_wrappers(oracle, dev):
    """The pybind-compatible entry points (int32, caller-allocated, CUDA-kernel semantics)."""
    from reart_amd import pointnet2_cuda as pc

    rng = np.random.default_rng(5)
    xyz = rng.uniform(-1, 1, (2, 777, 3)).astype(np.float32)
    pts = t(xyz, dev)
    idx = torch.zeros((2, 64), dtype=torch.int32, device=dev)
    temp = torch.full((2, 777), 1e10, device=dev)
    assert pc.furthest_point_sampling_wrapper(2, 777, 64, pts, temp, idx) == 1
    ref = oracle.fps(xyz, 64, start=None, cuda_mode=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    new = np.take_along_axis(xyz, ref[..., None].repeat(3, -1), 1)
    bidx = torch.zeros((2, 64, 16), dtype=torch.int32, device=dev)
    assert pc.ball_query_wrapper(2, 777, 64, 0.25, 16, t(new, dev), pts, bidx) == 1
    np.testing.assert_array_equal(bidx.cpu().numpy(), oracle.ball_query(0.25, 16, xyz, new, cuda_mode=True))
    # three_nn_wrapper (dead in the reference, interpolate_gpu.cu:81-131): the three nearest by direct-difference squared
    # distance = the oracle's K = 3 search
    d3 = torch.zeros((2, 777, 3), dtype=torch.float32, device=dev)
    i3 = torch.zeros((2, 777, 3), dtype=torch.int32, device=dev)
    pc.three_nn_wrapper(2, 777, 64, pts, t(new, dev), d3, i3)
    dd, ii = oracle.knn_points(xyz, new, K=3)
    np.testing.assert_array_equal(i3.cpu().numpy(), ii)
    np.testing.assert_array_equal(d3.cpu().numpy(), dd)
    # knn_wrapper (interpolate_gpu.cu:9-58): the same search with k columns
    d8 = torch.zeros((2, 777, 8), dtype=torch.float32, device=dev)
    i8 = torch.zeros((2, 777, 8), dtype=torch.int32, device=dev)
    pc.knn_wrapper(2, 777, 64, 8, pts, t(new, dev), d8, i8)
    dd, ii = oracle.knn_points(xyz, new, K=8)
    np.testing.assert_array_equal(i8.cpu().numpy(), ii)
    np.testing.assert_array_equal(d8.cpu().numpy(), dd)
    with pytest.raises(NotImplementedError):
        pc.knn_wrapper(2, 777, 64, 17, pts, t(new, dev), d8, i8)


@pytest.mark.parametrize("B,C,N,M,S", [(2, 5, 777, 64, 16), (1, 1, 1, 1, 1), (3, 128, 4096, 1024, 32), (2, 3, 50, 300, 3)])
def test_pointnet2_cuda_channel_major_operators(oracle, dev, B, C, N, M, S):
    """gather / group / three_interpolate and their backward forms (dead in the reference: pointnet2_api.cpp:14-25) against the
    oracle's restatements of the CUDA kernels: forward bit for bit, the atomically accumulated backward within 1e-5 relative
    of a float64 sum (repeated indices everywhere: M > N in the last case)."""
    from reart_amd import pointnet2_cuda as pc

    rng = np.random.default_rng(B * 1000 + C)
    feat = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, M)).astype(np.int32)
    out = torch.empty((B, C, M), device=dev)
    assert pc.gather_points_wrapper(B, C, N, M, t(feat, dev), t(idx, dev), out) == 1
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.pn2_gather_points(feat, idx))
    go = rng.normal(size=(B, C, M)).astype(np.float32)
    gp = torch.zeros((B, C, N), device=dev)
    assert pc.gather_points_grad_wrapper(B, C, N, M, t(go, dev), t(idx, dev), gp) == 1
    ref = oracle.pn2_gather_points_grad(go, idx, N)
    np.testing.assert_allclose(gp.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    # group: idx [B,M,S]
    gidx = rng.integers(0, N, size=(B, M, S)).astype(np.int32)
    gout = torch.empty((B, C, M, S), device=dev)
    assert pc.group_points_wrapper(B, C, N, M, S, t(feat, dev), t(gidx, dev), gout) == 1
    np.testing.assert_array_equal(gout.cpu().numpy(), oracle.pn2_gather_points(feat, gidx))
    ggo = rng.normal(size=(B, C, M, S)).astype(np.float32)
    ggp = torch.zeros((B, C, N), device=dev)
    assert pc.group_points_grad_wrapper(B, C, N, M, S, t(ggo, dev), t(gidx, dev), ggp) == 1
    ref = oracle.pn2_gather_points_grad(ggo, gidx, N)
    np.testing.assert_allclose(ggp.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(np.abs(ref).max(), 1.0))
    # interpolate: features known at N points, wanted at M (idx / weight [B,M,3])
    iidx = rng.integers(0, N, size=(B, M, 3)).astype(np.int32)
    w = rng.uniform(0, 1, size=(B, M, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    iout = torch.empty((B, C, M), device=dev)
    pc.three_interpolate_wrapper(B, C, N, M, t(feat, dev), t(iidx, dev), t(w, dev), iout)
    np.testing.assert_array_equal(iout.cpu().numpy(), oracle.pn2_three_interpolate(feat, iidx, w))
    igp = torch.zeros((B, C, N), device=dev)
    pc.three_interpolate_grad_wrapper(B, C, M, N, t(go, dev), t(iidx, dev), t(w, dev), igp)
    ref = oracle.pn2_three_interpolate_grad(go, iidx, w, N)
    np.testing.assert_allclose(igp.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(np.abs(ref).max(), 1.0))
    if C > 1 and N > 1:                                    # CHECK_CONTIGUOUS of the reference's wrappers
        with pytest.raises(RuntimeError):
            pc.gather_points_wrapper(B, N, C, M, t(feat, dev).transpose(1, 2), t(idx, dev), out)


def test_fps_full_size_properties(dev):
    """19 x 4096 -> 1024 (the assignment-loss shape): min-distance sequence is non-increasing,
    indices are distinct, first index is the injected start."""
    from reart_amd.networks.pointnet2_utils import farthest_point_sample, index_points

    gen = torch.Generator().manual_seed(0)
    xyz = (torch.rand((19, 4096, 3), generator=gen) - 0.5).to(dev)
    start = torch.arange(19, device=dev) * 7
    idx = farthest_point_sample(xyz, 1024, start=start)
    assert torch.equal(idx[:, 0], start)
    assert all(len(set(row.tolist())) == 1024 for row in idx.cpu())
    sel = index_points(xyz, idx)  # [19,1024,3]
    d = torch.cdist(sel[0], sel[0])
    prev = torch.tensor([d[i, :i].min() for i in range(1, 1024)])
    assert (prev[1:] <= prev[:-1] + 1e-6).all()
