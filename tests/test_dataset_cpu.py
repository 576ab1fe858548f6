"""reart_amd.dataset.Sequence and sparse_sample_novel_state (numpy mirrors of the reference's dataset/dataset_robot.py
and utils/dataset_utils.py) on the fixture directory tests/golden/seq_tiny against what the reference's own classes
returned for it (tests/golden/seq_tiny.npz, make_golden_dataset_ik.py): every array bit for bit."""
import os

import numpy as np

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "seq_tiny.npz"))


def test_sequence_sample_equals_the_reference():
    from reart_amd.dataset import Sequence
    from reart_amd.utils.dataset_utils import sparse_sample_novel_state

    seq = Sequence(os.path.join(HERE, "golden", "seq_tiny"), num_points=80, cano_idx=1)
    assert len(seq) == 1 and len(seq.pose_list) == 4 and len(seq.novel_pose_list) == 2
    sample = seq[0]
    for k in ("cano_pc", "gt_cano_part", "gt_flow_list", "gt_pc_list", "pc_list", "gt_pose_list", "complete_pc_list",
              "complete_gt_pc_list", "complete_gt_part_list"):
        assert sample[k].dtype == G[k].dtype, k
        np.testing.assert_array_equal(sample[k], G[k], err_msg=k)
    for s, novel in enumerate(seq.novel_pose_list):
        ns = sparse_sample_novel_state(sample["cano_pc"], sample["gt_cano_part"], seq.pose_list[seq.cano_idx], novel, 1)
        for k, v in ns.items():
            np.testing.assert_array_equal(v, G[f"novel{s}_{k}"], err_msg=k)
