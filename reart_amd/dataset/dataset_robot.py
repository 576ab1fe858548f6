"""Host-side mirror of the reference's robot sequence loader (``dataset/dataset_robot.py:9-100``): same directory
layout (``state_i.pkl`` = {pc, part_id}, ``pose_i.pkl`` / ``novel_pose_i.pkl`` = {part id: 4x4 pose relative to
state 0}), same attributes (``pose_list``, ``novel_pose_list``, ``cano_idx``) and the same sample dictionary.  Pure
numpy: this is input preparation, it runs once before anything touches the GPU."""
import glob
import os

import numpy as np

from ..utils.dataset_utils import get_rel_pose, load_pose, load_state, pose_identity_like


def _index_of(path):
    return int(os.path.basename(path).split(".")[0].split("_")[-1])


def _move_parts(points, part_ids, poses):
    """points [n,3] moved by the 4x4 pose of each point's part (float64 arithmetic like the reference, result in
    the dtype of ``points``)."""
    out = np.empty_like(points)
    for pid, pose in poses.items():
        sel = part_ids == pid
        homo = np.concatenate([points[sel], np.ones((int(sel.sum()), 1), dtype=float)], axis=1)
        out[sel] = (homo @ np.asarray(pose).T)[:, :3]
    return out


class Sequence(object):
    def __init__(self, seq_path, num_points=4096, cano_idx=0):
        self.seq_path = seq_path
        self.cat = seq_path.split("/")[-1]
        self.num_points = num_points
        self.cano_idx = cano_idx
        pose_files = sorted(glob.glob(os.path.join(seq_path, "pose_*.pkl")), key=_index_of)
        novel_files = sorted(glob.glob(os.path.join(seq_path, "novel_pose_*.pkl")), key=_index_of)
        self.pc_path_list = [os.path.join(seq_path, "state_0.pkl")]
        self.pc_path_list += [os.path.join(seq_path, "state_{}.pkl".format(_index_of(f))) for f in pose_files]
        self.pose_list = [load_pose(f) for f in pose_files]
        self.pose_list.insert(0, pose_identity_like(self.pose_list[0]))     # state 0 is the reference frame
        self.novel_pose_list = [load_pose(f) for f in novel_files]
        assert len(self.pc_path_list) == len(self.pose_list)

    def __len__(self):
        return 1

    def __getitem__(self, item):
        clouds, parts = [], []
        for path in self.pc_path_list:
            pc, part = load_state(path)
            clouds.append(pc[:self.num_points])
            parts.append(part[:self.num_points])
        complete_pc_list = np.stack(clouds).astype("float32")
        complete_gt_part_list = np.stack(parts)
        c = self.cano_idx
        cano_pc, gt_cano_part = complete_pc_list[c], complete_gt_part_list[c]
        part_ids = list(set(complete_gt_part_list[0].tolist()))
        moved, gt_pose_list = [], []
        for tgt_pose in self.pose_list:          # the canonical cloud carried to every frame by the ground-truth poses
            rel = get_rel_pose(self.pose_list[c], tgt_pose)
            moved.append(_move_parts(cano_pc, gt_cano_part, {p: rel[p] for p in part_ids}))
            gt_pose_list.append(np.stack([rel[p] for p in part_ids]).astype("float32"))
        complete_gt_pc_list = np.stack(moved).astype("float32")
        drop = lambda a: np.concatenate((a[:c], a[c + 1:]), axis=0)
        return {"cano_pc": cano_pc, "gt_cano_part": gt_cano_part,
                "gt_flow_list": complete_gt_pc_list[1:] - complete_gt_pc_list[:-1],
                "gt_pc_list": drop(complete_gt_pc_list), "pc_list": drop(complete_pc_list),
                "gt_pose_list": np.stack(gt_pose_list).astype("float32"), "complete_pc_list": complete_pc_list,
                "complete_gt_pc_list": complete_gt_pc_list, "complete_gt_part_list": complete_gt_part_list}
