"""GPU parity of the K-NN / Chamfer kernels against the CPU oracle (bit-exact indices and
distances), through the C ABI via the reference-shaped Python surface."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(rng, N, P1, P2, scale=0.35):
    a = rng.uniform(-scale, scale, (N, P1, 3)).astype(np.float32)
    b = rng.uniform(-scale, scale, (N, P2, 3)).astype(np.float32)
    return a, b


@pytest.mark.parametrize("N,P1,P2", [(1, 1, 1), (2, 64, 64), (3, 100, 257), (2, 513, 40), (4, 1024, 1000)])
@pytest.mark.parametrize("K", [1, 3])
def test_knn_points_bit_exact(oracle, dev, N, P1, P2, K):
    from reart_amd.utils.chamfer import knn_points

    if K > P2:
        pytest.skip("K > P2 covered by the ragged test")
    rng = np.random.default_rng(100 * N + P1 + K)
    a, b = _clouds(rng, N, P1, P2)
    d_ref, i_ref = oracle.knn_points(a, b, K=K)
    out = knn_points(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), K=K)
    assert out.idx.dtype == torch.int64 and out.dists.dtype == torch.float32
    np.testing.assert_array_equal(out.idx.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(out.dists.cpu().numpy(), d_ref)


def test_knn_points_ties_lowest_index(oracle, dev):
    """Duplicated targets: every tie must resolve to the lowest index, across slice borders."""
    from reart_amd.utils.chamfer import knn_points

    rng = np.random.default_rng(7)
    base = rng.uniform(-1, 1, (1, 300, 3)).astype(np.float32)
    b = np.concatenate([base, base, base], axis=1)  # each target three times
    a = rng.uniform(-1, 1, (1, 500, 3)).astype(np.float32)
    for K in (1, 4):
        d_ref, i_ref = oracle.knn_points(a, b, K=K)
        out = knn_points(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), K=K)
        np.testing.assert_array_equal(out.idx.cpu().numpy(), i_ref)
        np.testing.assert_array_equal(out.dists.cpu().numpy(), d_ref)
    assert (i_ref[..., 0] < 300).all()


def test_knn_points_ragged_lengths(oracle, dev):
    from reart_amd.utils.chamfer import knn_points

    rng = np.random.default_rng(3)
    a, b = _clouds(rng, 3, 200, 150)
    l1 = np.array([200, 17, 0], np.int64)
    l2 = np.array([150, 2, 90], np.int64)
    for K in (1, 3):
        d_ref, i_ref = oracle.knn_points(a, b, l1, l2, K=K)
        out = knn_points(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev),
                         lengths1=torch.from_numpy(l1).to(dev), lengths2=torch.from_numpy(l2).to(dev), K=K)
        np.testing.assert_array_equal(out.idx.cpu().numpy(), i_ref)
        np.testing.assert_array_equal(out.dists.cpu().numpy(), d_ref)


def test_chamfer_module_matches_oracle_and_grad(oracle, dev):
    from reart_amd.utils.chamfer import ChamferDistance

    rng = np.random.default_rng(11)
    x, y = _clouds(rng, 3, 700, 700)
    d1, i1, d2, i2 = oracle.chamfer_bidir(x, y)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    yt = torch.from_numpy(y).to(dev).requires_grad_(True)
    cd, fi, bi = ChamferDistance()(xt, yt, bidirectional=True, return_index=True)
    np.testing.assert_array_equal(fi.cpu().numpy(), i1)
    np.testing.assert_array_equal(bi.cpu().numpy(), i2)
    np.testing.assert_array_equal(cd.detach().cpu().numpy(), d1 + d2)
    cd.sum().backward()
    g = np.ones((3, 700, 1), np.float32)
    gx1, gy1 = oracle.knn_points_backward(x, y, i1[..., None], g)
    gy2, gx2 = oracle.knn_points_backward(y, x, i2[..., None], g)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gx1 + gx2, rtol=0, atol=1e-6)
    np.testing.assert_allclose(yt.grad.cpu().numpy(), gy1 + gy2, rtol=0, atol=1e-6)


def test_knn_backward_bit_exact_and_deterministic(oracle, dev):
    from reart_amd import chamferdist_C as C

    rng = np.random.default_rng(5)
    a, b = _clouds(rng, 2, 900, 300)  # many sources share one target: long buckets
    K = 2
    d, i = oracle.knn_points(a, b, K=K)
    g = rng.normal(size=d.shape).astype(np.float32)
    g1_ref, g2_ref = oracle.knn_points_backward(a, b, i, g)
    args = [torch.from_numpy(v).to(dev) for v in (a, b)]
    it, gt = torch.from_numpy(i).to(dev), torch.from_numpy(g).to(dev)
    g1, g2 = C.knn_points_backward(args[0], args[1], None, None, it, gt)
    g1b, g2b = C.knn_points_backward(args[0], args[1], None, None, it, gt)
    np.testing.assert_array_equal(g1.cpu().numpy(), g1_ref)
    np.testing.assert_array_equal(g2.cpu().numpy(), g2_ref)
    assert torch.equal(g1, g1b) and torch.equal(g2, g2b)


def test_knn_cuda_class(oracle, dev):
    from reart_amd.knn_cuda import KNN

    rng = np.random.default_rng(9)
    ref = rng.uniform(-0.3, 0.3, (2, 777, 3)).astype(np.float32)
    qry = rng.uniform(-0.3, 0.3, (2, 333, 3)).astype(np.float32)
    d_ref, i_ref = oracle.knn_cuda(ref, qry, 3, euclidean=True)
    knn = KNN(k=3, transpose_mode=True)
    assert knn.k == 3
    d, i = knn(torch.from_numpy(ref).to(dev), torch.from_numpy(qry).to(dev))
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
    np.testing.assert_allclose(d.cpu().numpy(), d_ref, rtol=2e-7, atol=0)  # sqrtf 1 ulp
    knn_t = KNN(k=1, transpose_mode=False)
    d, i = knn_t(torch.from_numpy(ref).to(dev).transpose(1, 2), torch.from_numpy(qry).to(dev).transpose(1, 2))
    assert tuple(i.shape) == (2, 1, 333)
    np.testing.assert_array_equal(i[:, 0].cpu().numpy(), i_ref[..., 0])


def test_chamfer_full_size_properties(dev):
    """BASELINE size (19 x 4096): size-independent properties instead of the O(N^2) oracle."""
    from reart_amd import chamferdist_C as C

    g = torch.Generator(device="cpu").manual_seed(2)
    x = (torch.rand((19, 4096, 3), generator=g) * 0.7 - 0.35).to(dev)
    y = (torch.rand((19, 4096, 3), generator=g) * 0.7 - 0.35).to(dev)
    d_xy, i_xy, d_yx, i_yx = C.chamfer_bidir(x, y)
    # 1. reported distance == distance to the reported index, recomputed in the same order
    nn = torch.gather(y, 1, i_xy[..., None].expand(-1, -1, 3))
    diff = x - nn
    rec = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert torch.equal(rec, d_xy)
    # 2. it is the minimum (and the lowest index attaining it) over all targets, on a query subsample
    for b in range(0, 19, 6):
        diff = x[b, ::8, None, :] - y[b, None, :, :]  # [512, 4096, 3]
        sq = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        mn, am = sq.min(dim=1)
        assert torch.equal(mn, d_xy[b, ::8]) and torch.equal(am, i_xy[b, ::8])
    # 3. self query: distance 0 at own index (ties -> lowest index == itself for distinct points)
    d_xx, i_xx, _, _ = C.chamfer_bidir(x, x)
    assert (d_xx == 0).all() and torch.equal(i_xx, torch.arange(4096, device=dev).expand(19, -1))
    # 4. permutation equivariance of the target set
    perm = torch.randperm(4096, generator=g).to(dev)
    d_p, i_p, _, _ = C.chamfer_bidir(x, y[:, perm])
    assert torch.equal(d_p, d_xy)


def test_knn_backward_degenerate_buckets(oracle, dev):
    """Every source picks the same target (P2 = 1 and a cluster): buckets of thousands of pairs
    must still be summed in ascending source order, bit-exact, in O(M) work."""
    from reart_amd import chamferdist_C as C

    rng = np.random.default_rng(6)
    for P2 in (1, 3):
        a = rng.uniform(-1, 1, (2, 5000, 3)).astype(np.float32)
        b = rng.uniform(-1, 1, (2, P2, 3)).astype(np.float32)
        d, i = oracle.knn_points(a, b, K=1)
        g = rng.normal(size=d.shape).astype(np.float32)
        g1_ref, g2_ref = oracle.knn_points_backward(a, b, i, g)
        g1, g2 = C.knn_points_backward(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), None, None,
                                       torch.from_numpy(i).to(dev), torch.from_numpy(g).to(dev))
        np.testing.assert_array_equal(g1.cpu().numpy(), g1_ref)
        np.testing.assert_array_equal(g2.cpu().numpy(), g2_ref)


def test_knn_per_point_against_the_reference_kdtree(dev):
    """HIP brute force and the fused loop's pruned warm-started search against the per-point results of the REFERENCE's
    KD-tree Chamfer (utils/eval_utils.py:39-66; tests/golden/chamfer_kdtree.npz) -- no oracle in between."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_golden_cpu import kdtree_check, load
    from reart_amd.utils.chamfer import knn_points

    g, c = load("chamfer_kdtree"), load("chamfer")

    def knn(p1, p2):
        out = knn_points(torch.from_numpy(np.ascontiguousarray(p1)).to(dev), torch.from_numpy(np.ascontiguousarray(p2)).to(dev), K=1)
        return out.dists[..., 0].cpu().numpy(), out.idx[..., 0].cpu().numpy()

    ties = (kdtree_check(knn, g, "ab", c["a"], c["b"]) + kdtree_check(knn, g, "st", c["src"], c["tgt"])
            + kdtree_check(knn, g, "nao", g["nao_x"], g["nao_y"]))
    assert ties <= 8

    # the pruned search of the fused step (k-d leaf order, boxes, seeds from a previous call)
    from reart_amd.chamferdist_C import knn_points_idx_warm
    from reart_amd.relax import kd_order

    def knn_w(p1, p2):
        """queries and targets stored in k-d leaf order like the engine stores them; results mapped back"""
        d_out, i_out = np.empty(p1.shape[:2], np.float32), np.empty(p1.shape[:2], np.int64)
        for b in range(p1.shape[0]):
            o1 = kd_order(torch.from_numpy(np.ascontiguousarray(p1[b]))).numpy()
            o2 = kd_order(torch.from_numpy(np.ascontiguousarray(p2[b]))).numpy()
            ta = torch.from_numpy(np.ascontiguousarray(p1[b][o1][None])).to(dev)
            tb = torch.from_numpy(np.ascontiguousarray(p2[b][o2][None])).to(dev)
            seeds = None
            for _ in range(2):      # cold, then warm from its own result
                idx, dists, seeds = knn_points_idx_warm(ta, tb, 1, seeds)
            d_out[b, o1] = dists[0, :, 0].cpu().numpy()
            i_out[b, o1] = o2[idx[0, :, 0].cpu().numpy()]
        return d_out, i_out

    kdtree_check(knn_w, g, "nao", g["nao_x"], g["nao_y"])
