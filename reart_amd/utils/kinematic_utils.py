"""Host-side mirror of ``fk`` from the reference's ``utils/kinematic_utils.py:151-198``."""
import numpy as np
import torch

from .. import _lib


def tree_arrays(edge_index, reverse_topo):
    """Flatten the reference's joint-tree dicts (edge_index: "child_parent" -> edge id,
    reverse_topo: parts from root to leaf; networks/model.py:79-93) into int32 arrays
    (parent, edge_of_part, order)."""
    P = len(reverse_topo)
    assert sorted(int(v) for v in reverse_topo) == list(range(P))
    parent = np.full(P, -1, np.int32)
    edge_of = np.full(P, -1, np.int32)
    for key, e in edge_index.items():
        c, p = (int(v) for v in key.split("_"))
        parent[c], edge_of[c] = p, int(e)
    order = np.asarray([int(v) for v in reverse_topo], np.int32)
    assert (parent < 0).sum() == 1, "the joint graph must be a tree with one root"
    return parent, edge_of, order


class _FK(torch.autograd.Function):
    """pc_trans, trans_list = rigid_apply(fk(...)); gradients to axis / moment / theta / distance."""

    @staticmethod
    def forward(ctx, x, part, axis, moment, theta, distance, parent, edge_of, order):
        _lib.require_gpu(x, part, axis, moment, theta)
        x, part = x.contiguous().float(), part.contiguous().long()
        axis, moment, theta = axis.contiguous().float(), moment.contiguous().float(), theta.contiguous().float()
        dist = None if distance is None else distance.contiguous().float()
        B, E = theta.shape
        P, N = parent.shape[0], x.shape[0]
        trans = torch.empty((B, P, 4, 4), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        rc = L.reart_fk_forward(_lib.ptr(parent), _lib.ptr(edge_of), _lib.ptr(order), P, _lib.ptr(axis),
                                _lib.ptr(moment), _lib.ptr(theta), _lib.ptr(dist), B, E, _lib.ptr(trans), _lib.stream())
        _lib.check(rc, "reart_fk_forward")
        out = torch.empty((B, N, 3), dtype=torch.float32, device=x.device)
        rc = L.reart_compute_pc_transform(_lib.ptr(x), _lib.ptr(trans), _lib.ptr(part), N, P, B, _lib.ptr(out),
                                          _lib.stream())
        _lib.check(rc, "reart_compute_pc_transform")
        ctx.save_for_backward(x, part, axis, moment, theta, trans, parent, edge_of, order)
        ctx.dist = dist
        ctx.mark_non_differentiable(trans)
        return out, trans

    @staticmethod
    def backward(ctx, g_out, g_trans):
        x, part, axis, moment, theta, trans, parent, edge_of, order = ctx.saved_tensors
        dist = ctx.dist
        B, E = theta.shape
        P, N = parent.shape[0], x.shape[0]
        G = g_out.contiguous().float()
        g_axis, g_moment, g_theta = torch.empty_like(axis), torch.empty_like(moment), torch.empty_like(theta)
        g_dist = None if dist is None else torch.empty_like(dist)
        L = _lib.lib()
        ws = _lib.workspace(L.reart_fk_backward_workspace_bytes(P, B, E), x.device)
        rc = L.reart_fk_backward(_lib.ptr(x), _lib.ptr(part), _lib.ptr(G), N, _lib.ptr(parent), _lib.ptr(edge_of),
                                 _lib.ptr(order), P, _lib.ptr(axis), _lib.ptr(moment), _lib.ptr(theta), _lib.ptr(dist),
                                 B, E, _lib.ptr(trans), _lib.ptr(g_axis), _lib.ptr(g_moment), _lib.ptr(g_theta),
                                 _lib.ptr(g_dist), _lib.ptr(ws), ws.numel(), _lib.stream())
        _lib.check(rc, "reart_fk_backward")
        return None, None, g_axis, g_moment, g_theta, g_dist, None, None, None


def fk(paths_to_base, reverse_topo, edge_index, axis_list, moment_list, theta_list, distance_list=None,
       joint_type_list=None):
    """Forward kinematics over screw joints -> [T, P, 4, 4] (utils/kinematic_utils.py:151-198).
    ``paths_to_base`` is accepted for signature compatibility; with parts visited root to leaf the
    reference's path walk always stops at the first edge, so the tree's parent links suffice."""
    dev = theta_list.device
    parent, edge_of, order = (torch.from_numpy(a).to(dev) for a in tree_arrays(edge_index, reverse_topo))
    theta, dist = _effective_joint_values(theta_list, distance_list, joint_type_list)
    dummy_x = torch.zeros((1, 3), device=dev)
    dummy_part = torch.zeros((1,), dtype=torch.long, device=dev)
    _, trans = _FK.apply(dummy_x, dummy_part, axis_list, moment_list, theta, dist, parent, edge_of, order)
    return trans


def _effective_joint_values(theta_list, distance_list, joint_type_list):
    """utils/kinematic_utils.py:174-186: prismatic joints run with theta = 1e-6 and their distance,
    revolute joints with their theta and distance = 1e-6 (also the default without a type list)."""
    if joint_type_list is None:
        return theta_list, distance_list
    pris = torch.tensor([jt == "prismatic" for jt in joint_type_list], device=theta_list.device)
    theta = torch.where(pris[None, :], torch.full_like(theta_list, 1e-6), theta_list)
    dist = torch.where(pris[None, :], distance_list, torch.full_like(distance_list, 1e-6))
    return theta, dist
