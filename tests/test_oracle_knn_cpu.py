"""CPU: the oracle's K-NN against a literal numpy restatement and against the reference's
independent KD-tree Chamfer formulation (utils/eval_utils.py:39-66, restated with scipy)."""
import numpy as np
from scipy.spatial import cKDTree


def _brute(a, b, K):
    N, P1, _ = a.shape
    d = np.empty((N, P1, K), np.float32)
    i = np.empty((N, P1, K), np.int64)
    for n in range(N):
        diff = a[n][:, None, :] - b[n][None, :, :]
        sq = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        order = np.argsort(sq, axis=1, kind="stable")[:, :K]
        i[n] = order
        d[n] = np.take_along_axis(sq, order, axis=1)
    return d, i


def test_oracle_knn_equals_literal_numpy(oracle):
    rng = np.random.default_rng(0)
    a = rng.uniform(-0.35, 0.35, (2, 300, 3)).astype(np.float32)
    b = rng.uniform(-0.35, 0.35, (2, 411, 3)).astype(np.float32)
    for K in (1, 3, 5):
        d, i = oracle.knn_points(a, b, K=K)
        d_ref, i_ref = _brute(a, b, K)
        np.testing.assert_array_equal(i, i_ref)
        np.testing.assert_array_equal(d, d_ref)


def test_oracle_chamfer_vs_kdtree(oracle):
    rng = np.random.default_rng(1)
    x = rng.uniform(-0.35, 0.35, (1, 2048, 3)).astype(np.float32)
    y = rng.uniform(-0.35, 0.35, (1, 2048, 3)).astype(np.float32)
    d1, i1, d2, i2 = oracle.chamfer_bidir(x, y)
    dist, ids = cKDTree(y[0].astype(np.float64)).query(x[0].astype(np.float64))
    np.testing.assert_allclose(d1[0], dist ** 2, rtol=1e-5, atol=1e-10)
    assert (ids == i1[0]).mean() > 0.999  # float64 vs float32 may reorder exact near-ties
    dist2, _ = cKDTree(x[0].astype(np.float64)).query(y[0].astype(np.float64))
    np.testing.assert_allclose(d2[0], dist2 ** 2, rtol=1e-5, atol=1e-10)


def test_oracle_backward_matches_autograd(oracle):
    import torch

    rng = np.random.default_rng(2)
    a = rng.normal(size=(2, 50, 3)).astype(np.float32)
    b = rng.normal(size=(2, 60, 3)).astype(np.float32)
    d, i = oracle.knn_points(a, b, K=2)
    g = rng.normal(size=d.shape).astype(np.float32)
    g1, g2 = oracle.knn_points_backward(a, b, i, g)
    at, bt = torch.tensor(a, requires_grad=True), torch.tensor(b, requires_grad=True)
    nn = torch.gather(bt[:, None].expand(-1, 50, -1, -1), 2, torch.tensor(i)[..., None].expand(-1, -1, -1, 3))
    sq = ((at[:, :, None, :] - nn) ** 2).sum(-1)
    (sq * torch.tensor(g)).sum().backward()
    np.testing.assert_allclose(g1, at.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g2, bt.grad.numpy(), rtol=1e-5, atol=1e-6)
