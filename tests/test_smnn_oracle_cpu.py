"""oracle.match_smnn (numpy restatement of the reference's SMNN matching) against tests/golden/smnn.npz, produced by
the reference's own utils/flow_utils.py:match_smnn (tests/golden/make_golden_smnn.py)."""
import os

import numpy as np
import pytest

import oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "smnn.npz"))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_smnn_matches_reference(tag):
    pairs, ratio, margin = oracle.match_smnn(G[f"d1_{tag}"], G[f"d2_{tag}"], 0.9)
    np.testing.assert_array_equal(pairs, G[f"idx_{tag}"])
    np.testing.assert_allclose(ratio, G[f"dists_{tag}"][:, 0], rtol=2e-5)
    assert abs(margin - float(G[f"margin_{tag}"])) < 1e-6
