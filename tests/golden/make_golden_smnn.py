#!/usr/bin/env python3
"""Golden vectors for the SMNN descriptor matching, produced by the reference's own
``utils/flow_utils.py:match_smnn`` imported here:   python tests/golden/make_golden_smnn.py  ->  tests/golden/smnn.npz
Descriptors are synthetic 64-d vectors (the reference's trained extractor weights are not shipped): two noisy copies of
a common set plus unrelated ones, so that some rows pass the ratio test, some fail it and some are not mutual."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stand-ins + reference on sys.path)

from utils.flow_utils import match_smnn  # noqa: E402


def main():
    out = {}
    for tag, (n1, n2, shared, seed) in {"a": (900, 900, 700, 0), "b": (300, 512, 250, 1)}.items():
        g = torch.Generator().manual_seed(seed)
        base = torch.randn((max(n1, shared), 64), generator=g)
        d1 = base[:n1] + 0.15 * torch.randn((n1, 64), generator=g)
        d2 = torch.cat([base[torch.randperm(base.shape[0], generator=g)[:shared]] + 0.15 * torch.randn((shared, 64), generator=g),
                        torch.randn((n2 - shared, 64), generator=g)])
        dists, idx = match_smnn(d1, d2, th=0.9)
        dm = torch.cdist(d1.double(), d2.double())
        v = torch.topk(dm, 2, dim=1, largest=False)[0]
        v2 = torch.topk(dm.t(), 2, dim=1, largest=False)[0]
        margin = min(float((v[:, 0] / v[:, 1] - 0.9).abs().min()), float((v2[:, 0] / v2[:, 1] - 0.9).abs().min()))
        print(tag, "matches", idx.shape[0], "of", n1, "| closest ratio to the threshold:", margin)
        out.update({f"d1_{tag}": d1, f"d2_{tag}": d2, f"idx_{tag}": idx, f"dists_{tag}": dists, f"margin_{tag}": margin})
    mg.save("smnn", **out)


if __name__ == "__main__":
    main()
