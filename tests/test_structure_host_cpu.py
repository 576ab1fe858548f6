"""Host graph bookkeeping of the structure extraction (reart_amd/utils/graph_utils.py contract_edges,
reart_amd/utils/kinematic_utils.py JointTree) against networkx, which the reference uses (through oracle/structure.py):
same contraction result, same edge / node / path orderings, on random trees."""
import numpy as np
import pytest

from oracle import structure as S
from reart_amd.utils.graph_utils import contract_edges
from reart_amd.utils.kinematic_utils import JointTree


def _random_tree(rng, n):
    labels = sorted(rng.choice(40, n, replace=False).tolist())
    perm = rng.permutation(n)
    edges = []
    for k in range(1, n):
        a, b = labels[perm[k]], labels[perm[rng.integers(0, k)]]
        edges.append([a, b] if rng.random() < 0.5 else [b, a])
    rng.shuffle(edges)
    return labels, [list(map(int, e)) for e in edges]


@pytest.mark.parametrize("seed", range(60))
def test_contract_edges_matches_networkx(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 15))
    _, edges = _random_tree(rng, n)
    cost = rng.uniform(0, 1, len(edges)).tolist()
    thr = float(rng.choice([0.0, 0.3, 0.6, 1.1]))
    ref_relabel, ref_rest = S.contract_edges(edges, cost, thr)
    relabel, rest = contract_edges(edges, cost, thr)
    assert relabel == ref_relabel
    assert rest == ref_rest


@pytest.mark.parametrize("seed", range(40))
def test_joint_tree_orderings_match_networkx(seed):
    import networkx as nx

    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 15))
    perm = rng.permutation(n)
    edges = []
    for k in range(1, n):
        a, b = int(perm[k]), int(perm[rng.integers(0, k)])
        edges.append([a, b] if rng.random() < 0.5 else [b, a])
    rng.shuffle(edges)
    trans = np.tile(np.eye(4, dtype=np.float32), (3, n, 1, 1))
    trans[:, :, :3, 3] = rng.normal(size=(3, n, 3)).astype(np.float32)
    ref = S.build_graph(np.asarray(edges), trans)
    tree = JointTree(edges, ref["root"])
    assert tree.edges == [tuple(e) for e in ref["edges"]]
    assert tree.nodes == ref["nodes"]
    assert tree.reverse_topo == ref["reverse_topo"]
    assert {k: list(v) for k, v in tree.paths_to_base.items()} == {k: list(v) for k, v in ref["paths_to_base"].items()}


def test_rand_index_table_is_sized_from_the_data():
    """utils/eval_utils.py:25-36 (Rand index).  ADVICE r05: the sync-free form used a fixed 128 x 128 table and silently
    dropped labels beyond it; now the caller sizes the table and a label outside it gives NaN, not a wrong index."""
    import torch

    from reart_amd.utils.eval_utils import eval_seg

    rng = np.random.default_rng(0)
    gt, pd = torch.from_numpy(rng.integers(0, 200, 500)), torch.from_numpy(rng.integers(0, 7, 500))
    same = (gt[:, None] == gt[None, :]) == (pd[:, None] == pd[None, :])                       # the reference's N x N form
    want = float(same.double().mean())
    assert abs(float(eval_seg(gt, pd)) - want) < 1e-6
    assert abs(float(eval_seg(gt, pd, as_tensor=True, num_labels=200)) - want) < 1e-6
    assert np.isnan(float(eval_seg(gt, pd, as_tensor=True, num_labels=128)))
