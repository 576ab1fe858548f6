"""CPU: the oracle's whole relaxation iteration reproduces the reference's 10-step loss
trajectory (reference BaseModel + recon_loss + torch Adam, noise injected; G11)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_trajectory_matches_reference(oracle):
    from oracle.step import RelaxOracle, tau_cosine

    g = np.load(os.path.join(G, "trajectory.npz"))
    B, P = 9, 20
    p6d = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1))
    pt = np.zeros((B, P, 3), np.float32)
    eng = RelaxOracle(g["cano"], g["pcs"], g["W1_0"], g["b1_0"], g["W2_0"], p6d, pt, cano_idx=2)
    for i in range(10):
        assert abs(tau_cosine(i + 1, 15000, 1, 5) - g["taus"][i]) < 1e-12
        out = eng.step(g["noises"][i])
        assert abs(out["recon"] - g["losses"][i]) <= 1e-4 * g["losses"][i], (i, out["recon"], g["losses"][i])
    for k, ref in (("W1", "W1_f"), ("b1", "b1_f"), ("W2", "W2_f"), ("p6d", "p6d_f"), ("pt", "pt_f")):
        np.testing.assert_allclose(eng.params[k], g[ref], rtol=0, atol=2e-4 * max(1.0, np.abs(g[ref]).max()))
