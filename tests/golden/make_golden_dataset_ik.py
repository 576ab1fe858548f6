#!/usr/bin/env python3
"""Fixtures for the input side and the retargeting metric, produced by IMPORTING the reference:
    python tests/golden/make_golden_dataset_ik.py
  tests/golden/seq_tiny/        a tiny sequence directory in the reference's on-disk format (written HERE from seeded
                                random parts and rigid motions: data, 3 parts x 4 states + 2 novel poses)
  tests/golden/seq_tiny.npz     what the reference's dataset/dataset_robot.py:Sequence and
                                utils/dataset_utils.py:sparse_sample_novel_state return for it
  tests/golden/ik_nao.npz       the reference's utils/kinematic_utils.py:ik on its demo sequence with its shipped
                                kinematic-2 checkpoint: per novel pose the sparse samples, the ground-truth novel cloud
                                and the retarget error it reaches (CPU, 200 Adam iterations)"""
import copy
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from dataset.dataset_robot import Sequence  # noqa: E402
from utils.dataset_utils import sparse_sample_novel_state  # noqa: E402
from utils.kinematic_utils import ik  # noqa: E402
from networks.model import KinematicModel  # noqa: E402
from knn_cuda import KNN  # noqa: E402


def rigid(rng, scale):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    ang = rng.uniform(-scale, scale)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    T[:3, 3] = rng.normal(0, 0.1, 3)
    return T


def main():
    rng = np.random.default_rng(31)
    d = os.path.join(HERE, "seq_tiny")
    os.makedirs(d, exist_ok=True)
    part_ids = [0, 2, 5]
    for s in range(4):
        part = np.repeat(part_ids, 30)[rng.permutation(90)] if s else np.repeat(part_ids, 30)
        pc = rng.uniform(-0.3, 0.3, (90, 3))
        with open(os.path.join(d, f"state_{s}.pkl"), "wb") as f:
            pickle.dump({"pc": pc, "part_id": part}, f)
        if s:
            with open(os.path.join(d, f"pose_{s}.pkl"), "wb") as f:
                pickle.dump({p: rigid(rng, 0.8) for p in part_ids}, f)
    for s in range(2):
        with open(os.path.join(d, f"novel_pose_{s}.pkl"), "wb") as f:
            pickle.dump({p: rigid(rng, 1.2) for p in part_ids}, f)
    seq = Sequence(d, num_points=80, cano_idx=1)
    sample = seq[0]
    out = {k: v for k, v in sample.items()}
    for s, novel in enumerate(seq.novel_pose_list):
        ns = sparse_sample_novel_state(sample["cano_pc"], sample["gt_cano_part"], seq.pose_list[seq.cano_idx], novel, 1)
        out.update({f"novel{s}_{k}": v for k, v in ns.items()})
    mg.save("seq_tiny", **out)

    # ---- ik on the demo sequence with the shipped kinematic-2 checkpoint
    nao = Sequence(os.path.join(mg.REF, "demo_data/data/nao"), num_points=4096, cano_idx=2)
    ck = torch.load(os.path.join(mg.REF, "demo_data/pretrained/nao/kinematic-2/model.pth.tar"), map_location="cpu",
                    weights_only=False)
    model = KinematicModel(pose_len=9, seg_part=ck["seg_part"], cano_pc=ck["cano_pc"], knn=KNN(k=1, transpose_mode=True),
                           edge_index=ck["edge_index"], paths_to_base=ck["paths_to_base"], reverse_topo=ck["reverse_topo"])
    model.load_state_dict(ck["state_dict"], strict=True)
    s0 = nao[0]
    res = {"mean_err": ik(nao, model, "cpu", verbose=False, vis=False)}
    errs = []
    for s, novel in enumerate(nao.novel_pose_list):
        one = copy.copy(nao)
        one.novel_pose_list = [novel]
        errs.append(ik(one, model, "cpu", verbose=False, vis=False))
        ns = sparse_sample_novel_state(s0["cano_pc"], s0["gt_cano_part"], nao.pose_list[nao.cano_idx], novel, 1)
        res.update({f"sparse_cano_{s}": ns["sparse_cano_pc"], f"sparse_novel_{s}": ns["sparse_novel_pc"],
                    f"novel_pc_{s}": ns["novel_pc"].astype(np.float32)})
    res["errs"] = np.asarray(errs)
    print("retarget errors", errs, "mean", res["mean_err"])
    mg.save("ik_nao", **res)


if __name__ == "__main__":
    main()
