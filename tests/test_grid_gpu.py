"""GPU: the grid-accelerated search returns exactly what the brute-force search returns
(distances and indices bit for bit, ties -> lowest index), on friendly and hostile inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grid_knn(dev, tgt, qry, K, offsets=None):
    from reart_amd import _lib

    L = _lib.lib()
    t = torch.from_numpy(np.ascontiguousarray(tgt)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(qry)).to(dev)
    E, nq = q.shape[0], q.shape[1]
    if offsets is None:
        nt, off = t.shape[1], None
    else:
        off = torch.from_numpy(np.asarray(offsets, np.int32)).to(dev)
        nt = int(np.diff(offsets).max())
    d = torch.empty((E, nq, K), device=dev)
    i = torch.empty((E, nq, K), dtype=torch.int32, device=dev)
    ws = torch.empty(L.reart_grid_knn_workspace_bytes(E, nt), dtype=torch.uint8, device=dev)
    rc = L.reart_grid_knn(_lib.ptr(t), _lib.ptr(off), E, nt, _lib.ptr(q), nq, K, _lib.ptr(d), _lib.ptr(i), _lib.ptr(ws),
                          ws.numel(), _lib.stream())
    _lib.check(rc, "reart_grid_knn")
    return d.cpu().numpy(), i.cpu().numpy()


def _cases(rng):
    yield "uniform", rng.uniform(-0.35, 0.35, (3, 4096, 3)), rng.uniform(-0.35, 0.35, (3, 1000, 3))
    surf = rng.normal(size=(2, 3000, 3)); surf /= np.linalg.norm(surf, axis=-1, keepdims=True)
    yield "sphere surface, queries inside/outside/far", surf * 0.3, np.concatenate(
        [rng.uniform(-0.05, 0.05, (2, 300, 3)), rng.uniform(-1, 1, (2, 300, 3)), rng.uniform(5, 9, (2, 50, 3))], 1)
    clus = np.concatenate([rng.normal(0, 1e-3, (1, 2000, 3)), rng.uniform(-1, 1, (1, 100, 3))], 1)
    yield "one tight cluster + outliers", clus, rng.uniform(-1.2, 1.2, (1, 800, 3))
    lat = np.stack(np.meshgrid(*[np.arange(12) * 0.05] * 3, indexing="ij"), -1).reshape(1, -1, 3)
    yield "lattice (exact distance ties everywhere)", lat, np.concatenate([lat[:, ::3] + 0.025, lat[:, ::5]], 1)
    dup = rng.uniform(-0.3, 0.3, (1, 500, 3))
    yield "every target three times", np.concatenate([dup, dup, dup], 1), rng.uniform(-0.3, 0.3, (1, 600, 3))
    yield "flat sheet (degenerate extent in z)", np.concatenate(
        [rng.uniform(-0.3, 0.3, (1, 2500, 2)), np.zeros((1, 2500, 1))], -1), rng.uniform(-0.4, 0.4, (1, 700, 3))
    yield "tiny set", rng.uniform(-1, 1, (2, 5, 3)), rng.uniform(-2, 2, (2, 100, 3))


@pytest.mark.parametrize("K", [1, 3])
def test_grid_equals_brute_force(oracle, dev, K):
    rng = np.random.default_rng(17)
    for name, tgt, qry in _cases(rng):
        tgt, qry = tgt.astype(np.float32), qry.astype(np.float32)
        d_ref, i_ref = oracle.knn_points(qry, tgt, K=K)
        d, i = _grid_knn(dev, tgt, qry, K)
        np.testing.assert_array_equal(i, i_ref, err_msg=name)
        np.testing.assert_array_equal(d, d_ref, err_msg=name)


def test_grid_ragged_sets(oracle, dev):
    rng = np.random.default_rng(4)
    lens = [3000, 17, 800, 2999]
    sets = [rng.uniform(-0.3, 0.3, (m, 3)).astype(np.float32) for m in lens]
    off = np.concatenate([[0], np.cumsum(lens)])
    qry = rng.uniform(-0.4, 0.4, (4, 512, 3)).astype(np.float32)
    d, i = _grid_knn(dev, np.concatenate(sets)[None], qry, 3, offsets=off)
    for e, s in enumerate(sets):
        d_ref, i_ref = oracle.knn_points(qry[e:e + 1], s[None], K=3)
        np.testing.assert_array_equal(i[e], i_ref[0])
        np.testing.assert_array_equal(d[e], d_ref[0])
