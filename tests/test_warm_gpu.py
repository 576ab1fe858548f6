"""The warm-started, box-pruned search (prune.hip: the kernels the fused step runs every iteration)
must return exactly what the brute-force contract returns -- indices and distances bit for bit --
for any seeds, any storage order, with ties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(oracle, dev, a, b, K, seed):
    from reart_amd.chamferdist_C import knn_points_idx_warm

    d_ref, i_ref = oracle.knn_points(a, b, K=K)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ts = None if seed is None else torch.from_numpy(seed.astype(np.int32)).to(dev)
    idx, dists, ts = knn_points_idx_warm(ta, tb, K, ts)
    np.testing.assert_array_equal(idx.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(dists.cpu().numpy(), d_ref)
    np.testing.assert_array_equal(ts.cpu().numpy(), i_ref.astype(np.int32))   # seeds of the next call
    return ts


@pytest.mark.parametrize("K", [1, 3])
@pytest.mark.parametrize("N,P1,P2", [(1, 1, 3), (2, 64, 64), (3, 100, 257), (2, 513, 40), (2, 1500, 1000), (1, 4096, 4096)])
def test_warm_random_clouds_any_seed(oracle, dev, N, P1, P2, K):
    """Unsorted uniform clouds (boxes are huge: nothing can be pruned) with cold, random and garbage seeds."""
    rng = np.random.default_rng(17 * N + P1 + P2 + K)
    a = rng.uniform(-0.4, 0.4, (N, P1, 3)).astype(np.float32)
    b = rng.uniform(-0.4, 0.4, (N, P2, 3)).astype(np.float32)
    _check(oracle, dev, a, b, K, None)
    _check(oracle, dev, a, b, K, rng.integers(0, P2, (N, P1, K)))
    garbage = rng.integers(-5, P2 + 5, (N, P1, K))
    garbage[:, ::3, :] = 0                      # repeated seeds inside a query (K = 3)
    _check(oracle, dev, a, b, K, garbage)


@pytest.mark.parametrize("K", [1, 3])
def test_warm_coherent_clouds_over_iterations(oracle, dev, K):
    """Spatially ordered clouds that move a little between calls: the situation of the relaxation loop.
    The seeds returned by one call start the next; every call must stay exact while most boxes are skipped."""
    from reart_amd.relax import kd_order
    from reart_amd.synthetic import make_sequence

    seq = make_sequence(T=4, n_parts=4, pts_per_part=512, seed=5, with_flow=False)
    frames = seq["complete"].astype(np.float32)
    tgt = frames[1][kd_order(torch.from_numpy(frames[1])).numpy()][None]
    q0 = frames[0][kd_order(torch.from_numpy(frames[0])).numpy()][None]
    rng = np.random.default_rng(1)
    seed = None
    for it in range(4):
        q = (q0 + rng.normal(0, 2e-3 * (it + 1), q0.shape)).astype(np.float32)
        ts = _check(oracle, dev, q, tgt, K, None if seed is None else seed.cpu().numpy())
        seed = ts


@pytest.mark.parametrize("K", [1, 3])
def test_warm_ties_lowest_index(oracle, dev, K):
    """Every target appears three times (in different boxes and slices): ties must go to the lowest index,
    also when the seed points at a later copy."""
    rng = np.random.default_rng(7)
    base = rng.uniform(-1, 1, (1, 400, 3)).astype(np.float32)
    b = np.concatenate([base, base, base], axis=1)
    a = np.concatenate([rng.uniform(-1, 1, (1, 300, 3)).astype(np.float32), base[:, :200]], axis=1)  # exact hits too
    _, i_ref = oracle.knn_points(a, b, K=K)
    later = i_ref.copy()
    later[..., 0] += 800                                    # the same point, highest copy
    later = np.minimum(later, b.shape[1] - 1)
    _check(oracle, dev, a, b, K, None)
    _check(oracle, dev, a, b, K, later if K == 1 else None)
    if K == 1:
        assert (i_ref[..., 0] < 400).all()
