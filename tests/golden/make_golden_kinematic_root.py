#!/usr/bin/env python3
"""Golden vectors for KinematicModel with root motion, joint types and distances (the variant the reference's
run_sapien.py / run_real.py construct, networks/model.py:113-166), produced by the reference's own class imported here:
    python tests/golden/make_golden_kinematic_root.py   ->   tests/golden/kinematic_root.npz
Tree, segmentation and joint parameters: the shipped kinematic-2 checkpoint; root motion, distances and joint types:
seeded synthetic values (the reference ships no SAPIEN / real checkpoint)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from networks.model import KinematicModel  # noqa: E402
from knn_cuda import KNN  # noqa: E402
import screw_se3  # noqa: E402


def main():
    rng = np.random.default_rng(21)
    ck = torch.load(os.path.join(mg.REF, "demo_data/pretrained/nao/kinematic-2/model.pth.tar"), map_location="cpu",
                    weights_only=False)
    sd = ck["state_dict"]
    T, E = sd["theta_list"].shape
    d6 = torch.tensor([[1, 0, 0, 0, 1, 0]], dtype=torch.float32).repeat(T, 1) + torch.from_numpy(rng.normal(0, 0.2, (T, 6)).astype(np.float32))
    root = torch.eye(4).repeat(T, 1, 1)
    root[:, :3, :3] = screw_se3.rotation_6d_to_matrix(d6)
    root[:, :3, 3] = torch.from_numpy(rng.normal(0, 0.05, (T, 3)).astype(np.float32))
    dist = torch.from_numpy(rng.normal(0, 0.03, (T, E)).astype(np.float32))
    types = ["prismatic" if e in (2, 5) else "revolute" for e in range(E)]
    model = KinematicModel(pose_len=T, seg_part=ck["seg_part"], cano_pc=ck["cano_pc"], knn=KNN(k=1, transpose_mode=True),
                           edge_index=ck["edge_index"], paths_to_base=ck["paths_to_base"], reverse_topo=ck["reverse_topo"],
                           axis_list=sd["axis_list"].clone(), moment_list=sd["moment_list"].clone(),
                           theta_list=sd["theta_list"].clone(), distance_list=dist.clone(), root_trans=root.clone(),
                           joint_type_list=types)
    x = ck["cano_pc"].float()[:768].contiguous()
    out, seg, trans = model(x)
    G = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32))
    (out * G).sum().backward()
    names = sorted(ck["edge_index"], key=ck["edge_index"].get)
    mg.save("kinematic_root", cano_pc=ck["cano_pc"].float(), seg_part=ck["seg_part"].long(), input_pc=x,
            axis=sd["axis_list"], moment=sd["moment_list"], theta=sd["theta_list"], distance=dist, root_trans=root,
            root_6d=model.root_6d.detach(), root_t=model.root_t.detach(),
            prismatic=np.array([t == "prismatic" for t in types]),
            edge_child=[int(n.split("_")[0]) for n in names], edge_parent=[int(n.split("_")[1]) for n in names],
            reverse_topo=np.asarray(ck["reverse_topo"]), out=out.detach(), seg=seg, trans=trans.detach(), G=G,
            g_axis=model.axis_list.grad, g_moment=model.moment_list.grad, g_theta=model.theta_list.grad,
            g_distance=model.distance_list.grad, g_root_6d=model.root_6d.grad, g_root_t=model.root_t.grad)


if __name__ == "__main__":
    main()
