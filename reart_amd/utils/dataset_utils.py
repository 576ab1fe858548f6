"""Host-side helpers mirrored from the reference's ``utils/dataset_utils.py`` (numpy: input preparation and the sparse
retargeting samples of ``ik``; nothing here touches the GPU)."""
import pickle

import numpy as np


def load_normalize_dict(normalize_file):
    """category -> {'centroid', 'scale'} (the reference pickles it next to the data; run_robot.py:72-75)."""
    with open(normalize_file, "rb") as f:
        return pickle.load(f)


def load_state(load_path):
    """state_i.pkl -> (points [n,3], part ids [n])  (utils/dataset_utils.py:15-20)."""
    with open(load_path, "rb") as f:
        info = pickle.load(f)
    return info["pc"], info["part_id"]


def load_pose(load_path):
    """pose_i.pkl -> {part id: 4x4 pose relative to state 0}  (utils/dataset_utils.py:23-26)."""
    with open(load_path, "rb") as f:
        return pickle.load(f)


def get_rel_pose(pose_cano2src, pose_cano2tgt):
    """Per part: T_tgt * T_src^-1  (utils/dataset_utils.py:35-39)."""
    return {p: pose_cano2tgt[p] @ np.linalg.inv(pose_cano2src[p]) for p in pose_cano2src.keys()}


def pose_identity_like(pose_dict):
    """utils/dataset_utils.py:48-52."""
    return {p: np.eye(4) for p in pose_dict.keys()}


def sparse_sample_novel_state(cano_pc, gt_cano_part, cano_pose, novel_pose, sparse_sample_per_part=1):
    """The retargeting sample of ``ik`` (utils/dataset_utils.py:55-88): the canonical cloud carried to a novel pose by
    the ground-truth part poses, and ``sparse_sample_per_part`` fixed points per part (the 11th, 12th, ... of each part)
    before / after the motion."""
    ids = sorted(set(np.asarray(gt_cano_part).tolist()))
    rel = get_rel_pose(cano_pose, novel_pose)
    k = sparse_sample_per_part
    novel_pc = np.empty_like(cano_pc)
    sparse0, sparse1, sparse_id, poses = np.empty((k * len(ids), 3)), np.empty((k * len(ids), 3)), np.empty(k * len(ids)), []
    for n, pid in enumerate(ids):
        pose = rel[pid]
        poses.append(pose)
        sel = gt_cano_part == pid
        pts = cano_pc[sel]
        assert len(pts) > 10 + k
        move = lambda x: (np.concatenate([x, np.ones((x.shape[0], 1), dtype=float)], axis=1) @ pose.T)[:, :3]
        novel_pc[sel] = move(pts)
        pick = pts[10:10 + k]
        sparse0[n * k:(n + 1) * k], sparse1[n * k:(n + 1) * k], sparse_id[n * k:(n + 1) * k] = pick, move(pick), pid
    return {"gt_novel_pose": np.stack(poses).astype("float32"), "gt_sparse_part": sparse_id, "novel_pc": novel_pc,
            "sparse_cano_pc": sparse0, "sparse_novel_pc": sparse1}
