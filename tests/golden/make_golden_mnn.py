#!/usr/bin/env python3
"""Golden vectors for the NON-default matching="mnn" branch of the reference's compute_corr_list_filter
(utils/flow_utils.py:116-143: k = 1 KNN on the descriptors in both directions + find_mutual_correspondences :102-113),
produced by calling the reference's own function:   python tests/golden/make_golden_mnn.py  ->  tests/golden/mnn.npz

The descriptors are those of tests/golden/smnn.npz (the trained extractor weights are not shipped): the "extractor" handed
to the reference returns them, so what the fixture pins is the matching itself.  `knn_cuda.KNN` is a third-party package
that is not vendored; the stand-in below is the plain definition (float64 distance matrix, k smallest per query) in the
package's non-transposed layout ([b, dim, n] -> indices [b, k, nq]).  The float64 gap between the best and second-best
descriptor distance is stored too: the match set is only well defined where that gap exceeds fp32 rounding."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stand-ins + reference on sys.path)

from utils.flow_utils import compute_corr_list_filter  # noqa: E402


class KNN1:
    """knn_cuda.KNN(k=1, transpose_mode=False) by its definition."""

    def __call__(self, ref, query):
        d = torch.cdist(query.double().transpose(1, 2), ref.double().transpose(1, 2))       # [b, nq, nr]
        v, i = torch.topk(d, 1, dim=2, largest=False)
        return v.transpose(1, 2).float(), i.transpose(1, 2)


def main():
    g = np.load(os.path.join(HERE, "smnn.npz"))
    out = {}
    for tag in ("a", "b"):
        d1, d2 = torch.from_numpy(g[f"d1_{tag}"]), torch.from_numpy(g[f"d2_{tag}"])
        n = min(d1.shape[0], d2.shape[0])          # the function matches consecutive FRAMES: equal point counts
        d1, d2 = d1[:n], d2[:n]
        frames = torch.stack([d1, d2])             # "descriptors" of a two-frame sequence, point-major

        def extractor(x, _frames=frames):          # x: [1, 3, N] slices of the normalised clouds (ignored)
            extractor.calls += 1
            return _frames[extractor.calls - 1:extractor.calls].transpose(1, 2)            # [1, 64, N] like PointNet2Msg2
        extractor.calls = 0
        src, tgt = compute_corr_list_filter(torch.zeros(2, n, 3), extractor, KNN1(), matching="mnn")
        dm = torch.cdist(d1.double(), d2.double())
        v = torch.topk(dm, 2, dim=1, largest=False)[0]
        v2 = torch.topk(dm.t(), 2, dim=1, largest=False)[0]
        gap = min(float((v[:, 1] - v[:, 0]).min()), float((v2[:, 1] - v2[:, 0]).min()))
        print(tag, "mutual matches", src[0].shape[0], "of", n, "| smallest best/second gap:", gap)
        out.update({f"n_{tag}": n, f"src_{tag}": src[0], f"tgt_{tag}": tgt[0], f"gap_{tag}": gap})
    mg.save("mnn", **out)


if __name__ == "__main__":
    main()
