"""KinematicModel with root motion, prismatic joints and distances (the reference's SAPIEN / real-scan variant,
networks/model.py:113-166) against tests/golden/kinematic_root.npz, produced by the reference's own class and autograd
(tests/golden/make_golden_kinematic_root.py): forward 1e-6, gradients 2e-4 relative to the largest entry."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "kinematic_root.npz"))


def test_root_motion_forward_and_gradients(dev):
    from reart_amd.knn_cuda import KNN
    from reart_amd.networks.model import KinematicModel
    from reart_amd.utils.kinematic_utils import JointTree

    t = lambda k, dt=None: torch.from_numpy(np.ascontiguousarray(G[k])).to(dev)
    edges = list(zip(G["edge_child"].tolist(), G["edge_parent"].tolist()))
    edge_index = {f"{c}_{p}": k for k, (c, p) in enumerate(edges)}
    tree = JointTree([list(e) for e in edges], int(G["reverse_topo"][0]))
    types = ["prismatic" if b else "revolute" for b in G["prismatic"]]
    model = KinematicModel(pose_len=9, seg_part=t("seg_part"), cano_pc=t("cano_pc"), knn=KNN(k=1, transpose_mode=True),
                           edge_index=edge_index, paths_to_base=tree.paths_to_base, reverse_topo=G["reverse_topo"].tolist(),
                           axis_list=t("axis"), moment_list=t("moment"), theta_list=t("theta"), distance_list=t("distance"),
                           root_trans=t("root_trans"), joint_type_list=types).to(dev)
    np.testing.assert_allclose(model.root_6d.detach().cpu().numpy(), G["root_6d"], atol=1e-7)
    out, seg, trans = model(t("input_pc"))
    np.testing.assert_array_equal(seg.cpu().numpy(), G["seg"])
    np.testing.assert_allclose(out.detach().cpu().numpy(), G["out"], atol=2e-6)
    np.testing.assert_allclose(trans.cpu().numpy(), G["trans"], atol=2e-6)
    (out * t("G")).sum().backward()
    for name, key in (("axis_list", "g_axis"), ("moment_list", "g_moment"), ("theta_list", "g_theta"),
                      ("distance_list", "g_distance"), ("root_6d", "g_root_6d"), ("root_t", "g_root_t")):
        got, ref = getattr(model, name).grad.cpu().numpy(), G[key]
        assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max(), np.abs(ref).max())
    # identity root motion by flag (evaluation-time construction, networks/model.py:119-120)
    m2 = KinematicModel(pose_len=9, seg_part=t("seg_part"), cano_pc=t("cano_pc"), knn=KNN(k=1, transpose_mode=True),
                        edge_index=edge_index, paths_to_base=tree.paths_to_base, reverse_topo=G["reverse_topo"].tolist(),
                        load_root_trans=True, load_distance=True).to(dev)
    assert {"root_6d", "root_t", "distance_list", "theta_list", "axis_list", "moment_list"} <= set(m2.state_dict())
