"""Data files of the package.  ``nao_demo.npz``: the reference's demo sequence (demo_data/data/nao: 10 frames x 4096 points,
cano_idx 2, with its ground-truth flows / parts) as arrays -- what ``bench.py --config nao`` / ``nao_recipe`` and the tools
run on.  Written by ``tools/make_nao_demo.py`` from the committed fixture ``tests/golden/structure.npz`` (itself generated
by ``tests/golden/make_golden_structure.py`` through the reference's loader)."""
import os

import numpy as np


def load_nao_demo():
    """-> dict of numpy arrays: cano [4096,3], pc_list [9,4096,3], cano_idx, complete_gt_pc_list [10,4096,3],
    gt_flow_list [9,4096,3], gt_cano_part [4096]."""
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "nao_demo.npz")) as g:
        return {k: g[k] for k in g.files}
