"""GPU parity of the projection model (fk + screw / SE(3) maps + hard-label apply) against the
oracle and the golden vectors from the reference's kinematic-2 checkpoint (forward AND autograd)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _dicts(k):
    edge_index = {f"{c}_{int(k['parent'][c])}": int(k["edge_of_part"][c]) for c in range(len(k["parent"]))
                  if k["parent"][c] >= 0}
    return edge_index, [int(v) for v in k["order"]]


def test_kinematic_model_golden_forward_backward(oracle, dev):
    from reart_amd.knn_cuda import KNN
    from reart_amd.networks.model import KinematicModel

    k = np.load(os.path.join(G, "kinematic.npz"))
    edge_index, topo = _dicts(k)
    model = KinematicModel(pose_len=9, seg_part=t(k["seg_part"], dev), cano_pc=t(k["cano_pc"], dev),
                           knn=KNN(k=1, transpose_mode=True), edge_index=edge_index, paths_to_base=None,
                           reverse_topo=topo, axis_list=t(k["axis"], dev), moment_list=t(k["moment"], dev),
                           theta_list=t(k["theta"], dev)).to(dev)
    assert set(model.state_dict().keys()) == {"axis_list", "moment_list", "theta_list"}  # reference ckpt keys
    out, seg, trans = model(t(k["input_pc"], dev))
    np.testing.assert_array_equal(seg.cpu().numpy(), k["seg"])
    np.testing.assert_allclose(trans.cpu().numpy(), k["trans"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out.detach().cpu().numpy(), k["out"], rtol=0, atol=1e-6)
    ref = oracle.fk(k["parent"], k["edge_of_part"], k["order"], k["axis"], k["moment"], k["theta"])
    np.testing.assert_allclose(trans.cpu().numpy(), ref, rtol=0, atol=5e-7)
    (out * t(k["G"], dev)).sum().backward()
    for got, name in ((model.axis_list.grad, "g_axis"), (model.moment_list.grad, "g_moment"),
                      (model.theta_list.grad, "g_theta")):
        np.testing.assert_allclose(got.cpu().numpy(), k[name], rtol=0, atol=2e-4 * np.abs(k[name]).max(), err_msg=name)


def _torch_fk(parent, edge_of, order, axis, moment, theta, dist):
    """plain PyTorch fp64 restatement of the same math, for autograd on random trees"""
    B, E = theta.shape
    P = len(parent)
    eye = torch.eye(4, dtype=theta.dtype).expand(B, 4, 4)
    F = [None] * P
    for c in order:
        if parent[c] < 0:
            F[c] = eye
            continue
        e = edge_of[c]
        l, m, th, d = axis[e], moment[e], theta[:, e], dist[:, e]
        q = torch.linalg.cross(l, m)
        v = torch.linalg.cross(q, l)[None] + (d / th)[:, None] * l[None]
        om, u = th[:, None] * l[None], th[:, None] * v
        n2 = (om * om).sum(-1)
        ph = torch.clamp(n2, 1e-4).sqrt()
        K = torch.zeros(B, 3, 3, dtype=theta.dtype)
        K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -om[:, 2], om[:, 1], om[:, 2], -om[:, 0], -om[:, 1], om[:, 0]
        K2 = K @ K
        I = torch.eye(3, dtype=theta.dtype)[None]
        R = I + (ph.sin() / ph)[:, None, None] * K + ((1 - ph.cos()) / ph ** 2)[:, None, None] * K2
        V = I + ((1 - ph.cos()) / ph ** 2)[:, None, None] * K + ((ph - ph.sin()) / ph ** 3)[:, None, None] * K2
        T = torch.zeros(B, 4, 4, dtype=theta.dtype)
        T[:, :3, :3], T[:, :3, 3], T[:, 3, 3] = R, (V @ u[:, :, None])[:, :, 0], 1.0
        F[c] = F[parent[c]] @ T
    return torch.stack(F, dim=1)


def test_fk_random_tree_vs_torch_autograd(dev):
    """Random 12-part tree incl. tiny rotations below the eps clamp; gradients of a random
    functional vs float64 autograd of a plain PyTorch restatement (tolerance 2e-4 relative)."""
    from reart_amd.utils.kinematic_utils import _FK

    rng = np.random.default_rng(3)
    P, B, N = 12, 7, 500
    parent = np.array([-1] + [int(rng.integers(0, c)) for c in range(1, P)], np.int32)
    edge_of = np.array([-1] + list(rng.permutation(P - 1)), np.int32)
    order = np.arange(P, dtype=np.int32)
    axis = rng.normal(size=(P - 1, 3)); axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    moment = rng.normal(0, 0.3, (P - 1, 3))
    theta = rng.uniform(-2.5, 2.5, (B, P - 1)); theta[0, :3] = 3e-3  # |w| below the 1e-2 clamp
    dist = rng.normal(0, 0.05, (B, P - 1))
    x = rng.uniform(-0.3, 0.3, (N, 3)); part = rng.integers(0, P, N)
    Gw = rng.normal(size=(B, N, 3))
    td = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    A, M, TH, D = td(axis), td(moment), td(theta), td(dist)
    F = _torch_fk(parent, edge_of, order, A, M, TH, D)
    Tn = F[:, torch.tensor(part)]
    out_ref = (Tn[:, :, :3, :3] @ torch.tensor(x)[None, :, :, None])[..., 0] + Tn[:, :, :3, 3]
    (out_ref * torch.tensor(Gw)).sum().backward()
    f32 = lambda a: t(np.asarray(a, np.float32), dev)
    a_, m_, th_, d_ = (f32(v).requires_grad_(True) for v in (axis, moment, theta, dist))
    out, trans = _FK.apply(f32(x), t(part, dev), a_, m_, th_, d_, t(parent, dev), t(edge_of, dev), t(order, dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=0, atol=5e-6)
    (out * f32(Gw)).sum().backward()
    for got, ref, name in ((a_.grad, A.grad, "axis"), (m_.grad, M.grad, "moment"), (th_.grad, TH.grad, "theta"),
                           (d_.grad, D.grad, "distance")):
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-4 * ref.abs().max().item(),
                                   err_msg=name)
