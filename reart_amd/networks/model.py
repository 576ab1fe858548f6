"""Host-side mirror of the reference's ``networks/model.py``: BaseModel (``:11-70``).

Same constructor, parameter names (checkpoints of the reference load unchanged), forward
signature ``model(cano_pc, tau=..., proposal_6d=..., proposal_t=...)`` and return triple
``(pc_trans_list [T-1,N,3], seg_part [N], trans_list [T-1,P,4,4])``.  The forward is ONE
fused HIP kernel (seg head -> Gumbel-softmax -> 6D -> rigid apply) and the backward three
small ones, instead of the reference's ~40 PyTorch ops over [(T-1)*P, N, 3] temporaries.
"""
import torch
import torch.nn as nn

from .. import _lib
from .blocks import MLPConv1d


class _BaseForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cano, W1, b1, W2, p6d, pt, gumbel, tau):
        _lib.require_gpu(cano, W1, b1, W2, p6d, pt, gumbel)
        cano = cano.contiguous().float()
        w1, bb1, w2 = W1.reshape(W1.shape[0], 3).contiguous(), b1.contiguous(), W2.reshape(W2.shape[0], -1).contiguous()
        p6, ptt = p6d.contiguous(), pt.contiguous()
        gumbel = gumbel.contiguous()
        N, (B, P), H = cano.shape[0], p6.shape[:2], w1.shape[0]
        dev = cano.device
        out = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        seg = torch.empty((N,), dtype=torch.int64, device=dev)
        trans = torch.empty((B, P, 4, 4), dtype=torch.float32, device=dev)
        yT = torch.empty((P, N), dtype=torch.float32, device=dev)
        hT = torch.empty((H, N), dtype=torch.float32, device=dev)
        hard = torch.empty((N,), dtype=torch.int32, device=dev)
        rc = _lib.lib().reart_base_forward(_lib.ptr(cano), N, P, B, _lib.ptr(w1), _lib.ptr(bb1), _lib.ptr(w2), H,
                                           _lib.ptr(p6), _lib.ptr(ptt), _lib.ptr(gumbel), float(tau),
                                           _lib.ptr(out), _lib.ptr(seg), _lib.ptr(trans), _lib.ptr(yT), _lib.ptr(hT),
                                           _lib.ptr(hard), _lib.stream())
        _lib.check(rc, "reart_base_forward")
        ctx.save_for_backward(cano, w1, bb1, w2, p6, ptt, yT, hT, hard)
        ctx.tau = float(tau)
        ctx.shapes = (W1.shape, W2.shape)
        ctx.mark_non_differentiable(seg)
        return out, seg, trans

    @staticmethod
    def backward(ctx, g_out, g_seg, g_trans):
        cano, w1, bb1, w2, p6, ptt, yT, hT, hard = ctx.saved_tensors
        N, (B, P), H = cano.shape[0], p6.shape[:2], w1.shape[0]
        G = g_out.contiguous().float()
        gW1, gb1, gW2 = torch.empty_like(w1), torch.empty_like(bb1), torch.empty_like(w2)
        g6d, gt = torch.empty_like(p6), torch.empty_like(ptt)
        L = _lib.lib()
        ws = _lib.workspace(L.reart_base_backward_workspace_bytes(N, P, B, H), cano.device)
        rc = L.reart_base_backward(_lib.ptr(cano), N, P, B, _lib.ptr(w1), _lib.ptr(bb1), _lib.ptr(w2), H,
                                   _lib.ptr(p6), _lib.ptr(ptt), _lib.ptr(yT), _lib.ptr(hT), _lib.ptr(hard), ctx.tau,
                                   _lib.ptr(G), _lib.ptr(gW1), _lib.ptr(gb1), _lib.ptr(gW2), _lib.ptr(g6d),
                                   _lib.ptr(gt), _lib.ptr(ws), ws.numel(), _lib.stream())
        _lib.check(rc, "reart_base_backward")
        if g_trans is not None and g_trans.abs().sum() != 0:
            raise NotImplementedError("gradient through trans_list is not used by the relaxation loop")
        s1, s2 = ctx.shapes
        return None, gW1.reshape(s1), gb1, gW2.reshape(s2), g6d, gt, None, None


def sample_gumbel(shape, device, dtype=torch.float32):
    """The noise F.gumbel_softmax draws (``-empty.exponential_().log()``), from torch's current
    generator on ``device`` -- so ``torch.manual_seed`` controls it like in the reference."""
    return -torch.empty(shape, dtype=dtype, device=device).exponential_().log()


class BaseModel(nn.Module):
    """Relaxation model, cf. networks/model.py:11-70."""

    def __init__(self, num_parts, pose_len, joint_trajectory=None, init_6d=None, init_t=None):
        super().__init__()
        if joint_trajectory is not None:
            raise NotImplementedError("joint_trajectory is never passed by the reference's drivers")
        self.num_parts, self.pose_len = num_parts, pose_len
        chain = torch.stack([torch.arange(num_parts - 1), torch.arange(num_parts - 1) + 1], dim=1)
        self.register_buffer("joint_connection", chain.long())
        self.seg_head = MLPConv1d(3, (128, num_parts), bn=False, gn=False, last_activation="none")
        self.joint_trajectory = None
        if init_6d is None:
            ident = torch.tensor([[[1.0, 0, 0, 0, 1, 0]]]).repeat(pose_len, num_parts, 1)
            self.proposal_6d = nn.Parameter(ident, requires_grad=True)
        else:
            self.proposal_6d = nn.Parameter(init_6d, requires_grad=False)
        if init_t is None:
            self.proposal_t = nn.Parameter(torch.zeros(pose_len, num_parts, 3), requires_grad=True)
        else:
            self.proposal_t = nn.Parameter(init_t, requires_grad=False)

    def _weights(self):
        c1, c2 = self.seg_head.model[0], self.seg_head.model[2]
        return c1.weight, c1.bias, c2.weight

    def seg_forward(self, cano_pc, **kwargs):
        """Noise-free logits [N,P] (or their arg-max), cf. networks/model.py:33-37."""
        seg = self.seg_head(cano_pc.permute(1, 0).unsqueeze(0)).squeeze(0).permute(1, 0)
        return seg.argmax(dim=-1) if kwargs.get("argmax") else seg

    def forward(self, cano_pc, **kwargs):
        tau = kwargs.get("tau", 1.0)
        p6d = kwargs.get("proposal_6d", self.proposal_6d)
        pt = kwargs.get("proposal_t", self.proposal_t)
        gumbel = kwargs.get("gumbel")  # extension: injected noise (tests); default = torch RNG
        if gumbel is None:
            gumbel = sample_gumbel((cano_pc.shape[0], self.num_parts), cano_pc.device)
        W1, b1, W2 = self._weights()
        return _BaseForward.apply(cano_pc, W1, b1, W2, p6d, pt, gumbel, tau)


class KinematicModel(nn.Module):
    """Projection model over a screw-joint tree, cf. networks/model.py:73-166.

    Same constructor keywords and parameter names as the reference (its checkpoints load with
    ``strict=True``): ``axis_list`` [E,3], ``moment_list`` [E,3], ``theta_list`` [T-1,E] and the
    optional ``distance_list``.  forward = k-NN label transfer (utils/model_utils.py:41-51) +
    fused FK kernel + hard-label rigid apply.  Root motion (``root_trans`` / ``load_root_trans``, the SAPIEN / real-scan
    variant, networks/model.py:113-120,153-158) is a per-frame rigid transform composed after the kernels (a [T-1,3,3]
    Gram-Schmidt and one batched multiply in PyTorch, differentiable w.r.t. ``root_6d`` / ``root_t``)."""

    def __init__(self, pose_len, seg_part, cano_pc, knn, **kwargs):
        super().__init__()
        from ..utils.kinematic_utils import tree_arrays

        self.seg_part = seg_part.long()
        self.cano_pc = cano_pc
        self.pose_len = pose_len
        self.num_parts = len(torch.unique(self.seg_part))
        self.knn = knn
        if knn is not None:
            assert self.knn.k == 1
        self.edge_index = kwargs["edge_index"]
        self.paths_to_base = kwargs["paths_to_base"]
        self.reverse_topo = kwargs["reverse_topo"]
        E = len(self.edge_index)
        assert self.num_parts == E + 1  # P = E + 1
        if "root_trans" in kwargs:
            from ..screw_se3 import matrix_to_rotation_6d

            self.root_6d = nn.Parameter(matrix_to_rotation_6d(kwargs["root_trans"][:, :3, :3]).float(), requires_grad=True)
            self.root_t = nn.Parameter(kwargs["root_trans"][:, :3, 3].clone().float(), requires_grad=True)
        elif kwargs.get("load_root_trans"):
            self.root_6d = nn.Parameter(torch.tensor([[1.0, 0, 0, 0, 1, 0]]).repeat((pose_len, 1)), requires_grad=True)
            self.root_t = nn.Parameter(torch.zeros(pose_len, 3), requires_grad=True)
        for name, shape in (("axis_list", (E, 3)), ("moment_list", (E, 3)), ("theta_list", (pose_len, E))):
            init = kwargs[name] if name in kwargs else torch.zeros(shape)
            setattr(self, name, nn.Parameter(init.clone().float(), requires_grad=True))
        if "distance_list" in kwargs:
            self.distance_list = nn.Parameter(kwargs["distance_list"], requires_grad=True)
        elif kwargs.get("load_distance"):
            self.distance_list = nn.Parameter(torch.zeros(pose_len, E), requires_grad=True)
        self.joint_type_list = kwargs.get("joint_type_list")
        self._tree_np = tree_arrays(self.edge_index, self.reverse_topo)
        self._tree_dev = {}

    def _tree(self, device):
        if device not in self._tree_dev:
            self._tree_dev[device] = tuple(torch.from_numpy(a).to(device) for a in self._tree_np)
        return self._tree_dev[device]

    def seg_forward(self, input_pc, **kwargs):
        from ..utils.model_utils import knn_query

        return knn_query(input_pc, self.cano_pc, self.seg_part, self.knn)

    def forward(self, input_pc, **kwargs):
        from ..utils.kinematic_utils import _FK, _effective_joint_values

        seg_part = self.seg_forward(input_pc)
        theta_list = kwargs.get("theta_list", self.theta_list)
        distance_list = self.distance_list if hasattr(self, "distance_list") else None
        theta, dist = _effective_joint_values(theta_list, distance_list, self.joint_type_list)
        parent, edge_of, order = self._tree(input_pc.device)
        pc_trans_list, trans_list = _FK.apply(input_pc, seg_part, self.axis_list, self.moment_list, theta, dist,
                                              parent, edge_of, order)
        if hasattr(self, "root_6d") and hasattr(self, "root_t"):     # networks/model.py:153-158
            a1, a2 = self.root_6d[:, :3], self.root_6d[:, 3:]
            b1 = torch.nn.functional.normalize(a1, dim=-1)
            b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
            R = torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)                       # [T-1,3,3]
            pc_trans_list = torch.matmul(pc_trans_list, R.transpose(1, 2)) + self.root_t[:, None, :]
            with torch.no_grad():
                root = torch.zeros((R.shape[0], 4, 4), dtype=R.dtype, device=R.device)
                root[:, :3, :3], root[:, :3, 3], root[:, 3, 3] = R, self.root_t, 1.0
                trans_list = torch.matmul(root[:, None], trans_list)
        return pc_trans_list, seg_part, trans_list
