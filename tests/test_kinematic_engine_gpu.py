"""The autograd-free kinematic projection loop (reart_amd.kinematic_engine.KinematicEngine: operator calls + the
hand-derived FK backward + reart_adam_step, in place) against the same iterations through PyTorch autograd and
torch.optim.Adam (run_robot.OperatorLoop, the reference's own loop structure, run_robot.py:154-221)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _model(dev, k, cano):
    from reart_amd.knn_cuda import KNN
    from reart_amd.networks.model import KinematicModel

    edge_index = {f"{c}_{int(k['parent'][c])}": int(k["edge_of_part"][c]) for c in range(len(k["parent"])) if k["parent"][c] >= 0}
    topo = [int(v) for v in k["order"]]
    seg = t(k["seg_part"], dev)
    return KinematicModel(pose_len=9, seg_part=seg, cano_pc=cano, knn=KNN(k=1, transpose_mode=True), edge_index=edge_index,
                          paths_to_base=None, reverse_topo=topo, axis_list=t(k["axis"], dev), moment_list=t(k["moment"], dev),
                          theta_list=t(k["theta"], dev)).to(dev)


@pytest.mark.parametrize("with_flow,gap,wd", [(True, 1, 0.0), (False, 2, 0.0), (True, 2, 0.01)])
def test_engine_equals_the_autograd_loop(dev, with_flow, gap, wd):
    """kinematic-2 checkpoint of the reference (golden kinematic.npz) on its own canonical cloud: six iterations of the
    assignment (+ flow) branch; parameters after every step and all losses agree with the autograd loop."""
    from reart_amd import run_robot as rr
    from reart_amd.kinematic_engine import KinematicEngine

    k = np.load(os.path.join(G, "kinematic.npz"))
    cano = t(k["cano_pc"], dev)
    rng = np.random.default_rng(3)
    B, N = 9, cano.shape[0]
    with torch.no_grad():
        pcs = _model(dev, k, cano)(cano)[0]
    pcs = (pcs + t(rng.normal(0, 0.004, (B, N, 3)).astype(np.float32), dev)).contiguous()
    pcs = torch.stack([p[torch.from_numpy(rng.permutation(N)).to(dev)] for p in pcs])
    refs = flows = None
    if with_flow:
        comp = torch.cat((pcs[:2], cano[None], pcs[2:]), dim=0)
        sel = [torch.from_numpy(rng.permutation(N)[:300 + 7 * f]).to(dev) for f in range(B)]
        refs = [comp[f][s] for f, s in enumerate(sel)]
        flows = [(comp[f + 1][s] - comp[f][s]) * 0.5 for f, s in enumerate(sel)]
    argv = ["--model", "kinematic", "--use_assign_loss", "--assign_iter", "0", "--downsample", "4", "--assign_gap", str(gap),
            "--cano_idx", "2", "--weight_decay", str(wd)] + (["--use_flow_loss"] if with_flow else [])
    a = rr.build_parser().parse_args(argv)
    m_ref, m_eng = _model(dev, k, cano), _model(dev, k, cano)
    loop = rr.OperatorLoop(a, m_ref, cano, pcs, refs, flows)
    eng = KinematicEngine(m_eng, cano, pcs, 2, refs, flows, trans_lr=a.trans_lr, weight_decay=wd, assign_iter=0, assign_gap=gap,
                          downsample=4, lambda_assign=a.lambda_assign, lambda_flow=a.lambda_flow)
    for i in range(6):
        l_ref = loop.iteration(i)
        l_eng = eng.iteration(i)
        for key in l_ref:
            a_, b_ = float(l_ref[key].detach()), float(l_eng[key].detach())
            assert abs(a_ - b_) <= 2e-5 * abs(a_) + 1e-7, (i, key)
        for name in ("axis_list", "moment_list", "theta_list"):
            np.testing.assert_allclose(getattr(m_eng, name).detach().cpu().numpy(), getattr(m_ref, name).detach().cpu().numpy(),
                                       rtol=0, atol=2e-6, err_msg=f"iteration {i} {name}")
    assert eng.lap_solves == loop.lap_solves
