"""Retargeting (reart_amd.utils.kinematic_utils.ik_single, the body of the reference's ik, utils/kinematic_utils.py:
200-266) on the demo sequence with the shipped kinematic-2 parameters against the errors the reference's own ik reaches
(tests/golden/ik_nao.npz).  200 Adam(amsgrad) steps on 14 points are deterministic but not bit-reproducible across
devices; both land in the same optimum: 1 % relative."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


def test_retarget_error_matches_the_reference(dev):
    from reart_amd.knn_cuda import KNN
    from reart_amd.networks.model import KinematicModel
    from reart_amd.utils.kinematic_utils import JointTree, ik_single

    K, G = np.load(os.path.join(HERE, "golden", "kinematic.npz")), np.load(os.path.join(HERE, "golden", "ik_nao.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    parent, edge_of = K["parent"], K["edge_of_part"]
    edges = sorted(((int(edge_of[c]), c, int(parent[c])) for c in range(len(parent)) if parent[c] >= 0))
    edge_index = {f"{c}_{p}": e for e, c, p in edges}
    tree = JointTree([[c, p] for _, c, p in edges], int(K["order"][0]))
    model = KinematicModel(pose_len=9, seg_part=t(K["seg_part"]), cano_pc=t(K["cano_pc"]), knn=KNN(k=1, transpose_mode=True),
                           edge_index=edge_index, paths_to_base=tree.paths_to_base, reverse_topo=K["order"].tolist(),
                           axis_list=t(K["axis"]), moment_list=t(K["moment"]), theta_list=t(K["theta"])).to(dev)
    errs = []
    for s in range(3):
        novel = dict(sparse_cano_pc=G[f"sparse_cano_{s}"], sparse_novel_pc=G[f"sparse_novel_{s}"], novel_pc=G[f"novel_pc_{s}"])
        err, kw = ik_single(model, t(K["cano_pc"]), novel, dev)
        assert kw["theta_list"].shape == (1, 9)
        errs.append(err)
    np.testing.assert_allclose(errs, G["errs"], rtol=1e-2)
    assert abs(np.mean(errs) - float(G["mean_err"])) <= 1e-2 * float(G["mean_err"])
