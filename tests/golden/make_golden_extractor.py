#!/usr/bin/env python3
"""Golden vectors for the PointNet++ correspondence extractor (G9): the REFERENCE's PointNet2Msg2
(networks/feature_extractor.py:10-49, CPU fallbacks for FPS / ball query) with seeded random
weights (corr_model.pth.tar is not shipped) on a 1024-point nao cloud.  The weights are NOT stored:
tests regenerate them from the same numpy seed with `reart_amd.synthetic.extractor_state`.

    python tests/golden/make_golden_extractor.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REART_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


from reart_amd.synthetic import extractor_state  # noqa: E402  (seeded weights: corr_model.pth.tar is not shipped)


def main():
    sys.path.insert(0, REF)
    import tests.golden.make_golden as mg  # installs the stand-ins, imports the reference  # noqa: F401
    from networks.feature_extractor import PointNet2Msg2
    from dataset.dataset_robot import Sequence

    sample = Sequence(os.path.join(REF, "demo_data/data/nao"), num_points=4096, cano_idx=2)[0]
    pts = torch.from_numpy(sample["complete_pc_list"][[0, 5]][:, :1024]).float()  # [2,1024,3]
    pts = pts - pts.mean(dim=1, keepdim=True)
    pts = pts / pts.norm(dim=-1).max()                                            # unit-ish scale: radii 0.05..0.4
    xyz = pts.permute(0, 2, 1).contiguous()
    model = PointNet2Msg2(out_dim=64)
    model.load_state_dict(extractor_state(model))
    model.eval()
    torch.manual_seed(21)
    s1 = torch.randint(0, 1024, (2,), dtype=torch.long)
    s2 = torch.randint(0, 512, (2,), dtype=torch.long)
    torch.manual_seed(21)
    with torch.no_grad():
        l1_xyz, l1 = model.sa1(xyz, xyz)
        l2_xyz, l2 = model.sa2(l1_xyz, l1)
        torch.manual_seed(21)
        feat = model(xyz)
    np.savez_compressed(os.path.join(HERE, "extractor.npz"), xyz=xyz.numpy(), start1=s1.numpy(), start2=s2.numpy(),
                        l1_xyz=l1_xyz.numpy(), l1_points=l1.numpy(), l2_xyz=l2_xyz.numpy(), l2_points=l2.numpy(),
                        feat=feat.numpy())
    print("wrote extractor.npz", feat.shape, float(feat.abs().mean()), float(feat.abs().max()))


if __name__ == "__main__":
    main()
