"""Kinematic projection loop without autograd: the reference's loop body (run_robot.py:154-221) for ``--model kinematic``
(networks/model.py:137-166, utils/kinematic_utils.py:151-198) as a fixed sequence of operator calls on the launch
stream -- forward kinematics + rigid apply, assignment re-solve, assignment / flow gradients, the hand-derived FK backward
and Adam (``reart_adam_step``) on the model's own parameter tensors, in place.  ``run_robot.OperatorLoop`` is the same
iteration through PyTorch autograd and ``torch.optim.Adam``; ``tests/test_kinematic_engine_gpu.py`` holds the two together.

What an iteration costs the host here: a dozen ctypes calls and, when the assignment is refreshed, the copy of its result.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .networks.pointnet2_utils import farthest_point_sample, index_points
from .utils.flow_utils import blend_anchor_motion
from .utils.kinematic_utils import _effective_joint_values


class KinematicEngine:
    """State of one kinematic projection on one GPU.

    ``model``: a ``KinematicModel`` (any joint types; with or without root motion, networks/model.py:113-166) whose ``axis_list`` /
    ``moment_list`` / ``theta_list`` (and ``distance_list``, ``root_6d``, ``root_t``) are optimised IN PLACE; ``cano_pc`` [N,3], ``pc_list`` [T-1,N,3]; ``pc_ref_list`` / ``flow_ref_list``:
    the flow references of run_robot.py:81-84 or None.  Hyper-parameters carry the reference's flag names
    (run_robot.py:362-420): ``trans_lr`` (Adam lr of every kinematic parameter, :150-151), ``weight_decay``,
    ``use_assign_loss`` / ``assign_iter`` / ``assign_gap`` / ``downsample`` / ``lambda_assign``, ``lambda_flow``,
    ``use_robust_loss``.  Before ``assign_iter`` (or with ``use_assign_loss=False``) the iteration takes the Chamfer branch
    (run_robot.py:187-190): two exact K = 1 searches and ``knn_points_backward`` for dL/d pc_trans, no autograd there either.
    """

    def __init__(self, model, cano_pc, pc_list, cano_idx, pc_ref_list=None, flow_ref_list=None, trans_lr=1e-2,
                 weight_decay=0.0, assign_iter=0, assign_gap=5, downsample=4, lambda_assign=3e-1, lambda_flow=1.0,
                 use_robust_loss=False, smooth_weight=1e-2, knn_squared=False, use_assign_loss=True, src_idx=None, tgt_idx=None):
        _lib.require_gpu(cano_pc, pc_list)
        self.root = hasattr(model, "root_6d") and hasattr(model, "root_t")      # networks/model.py:153-158
        self._pris = None
        if model.joint_type_list is not None:
            # utils/kinematic_utils.py:174-186: prismatic joints run with theta = 1e-6 and their distance, revolute joints with
            # their theta and distance = 1e-6; the masked entries take no gradient (torch.where's backward in the autograd loop)
            if not hasattr(model, "distance_list"):
                raise NotImplementedError("joint types without distance_list")
            self._pris = torch.tensor([jt == "prismatic" for jt in model.joint_type_list], device=cano_pc.device)[None, :]
        self.model, self.dev = model, cano_pc.device
        self.cano = cano_pc.contiguous().float()
        self.pc_list = pc_list.contiguous().float()
        self.cano_idx = int(cano_idx)
        self.B, self.N = self.pc_list.shape[:2]
        self.lr, self.wd = float(trans_lr), float(weight_decay)
        self.assign_iter, self.assign_gap, self.lambda_assign = int(assign_iter), int(assign_gap), float(lambda_assign)
        self.use_assign = bool(use_assign_loss)
        self.lambda_flow, self.robust, self.smooth = float(lambda_flow), bool(use_robust_loss), float(smooth_weight)
        self.euclid = 0 if knn_squared else 1
        self.refs = None
        if pc_ref_list is not None:
            self.refs = [(r.reshape(-1, 3).contiguous().float(), f.reshape(-1, 3).contiguous().float())
                         for r, f in zip(pc_ref_list, flow_ref_list)]
            assert len(self.refs) == self.B
        with torch.no_grad():
            self.part = model.seg_forward(self.cano).contiguous().long()      # fixed: label transfer from the model's own cloud
        self.parent, self.edge_of, self.order = model._tree(self.dev)
        self.P = int(self.parent.shape[0])
        # parameters in the order torch.optim.Adam would see them (model.parameters())
        self.params = [p for p in model.parameters() if p.requires_grad]
        for p in self.params:
            assert p.is_cuda and p.is_contiguous() and p.dtype == torch.float32
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.grads = {id(p): torch.zeros_like(p) for p in self.params}
        self.lap_steps_log = []
        self.step_count = 0
        # run_robot.py:167-169: both FPS calls sample fixed clouds (start 0 on the reference's CUDA path): once
        # (``src_idx`` [n] / ``tgt_idx`` [T-1,n]: samples handed in instead -- the reference's CPU sampler starts from a random point,
        # networks/pointnet2_utils.py:90, so a trajectory the reference produced on a CPU comes with its samples:
        # tests/golden/kinematic_loop.npz)
        num_fps = self.N // int(downsample)
        zero = torch.zeros(1, dtype=torch.long, device=self.dev)
        self.src_idx = (farthest_point_sample(self.cano[None], num_fps, start=zero, cuda_mode=True)[0] if src_idx is None
                        else torch.as_tensor(src_idx, dtype=torch.long, device=self.dev).reshape(-1))          # [n]
        tgt_idx = (farthest_point_sample(self.pc_list, num_fps, start=zero.expand(self.B), cuda_mode=True) if tgt_idx is None
                   else torch.as_tensor(tgt_idx, dtype=torch.long, device=self.dev).reshape(self.B, -1))
        # the sampled targets are the COLUMNS of every re-solve of the run: numbered along a Z-order curve (lap.spatial_order)
        from .utils.lap import spatial_order
        self.tgt_order = spatial_order(index_points(self.pc_list, tgt_idx))                                     # [B,n] into the FPS order
        self.tgt_idx = tgt_idx.gather(1, self.tgt_order)
        self.tgt_pts = index_points(self.pc_list, self.tgt_idx).contiguous()                                    # [B,n,3]
        self.matched = None
        self.lap_state = {}
        self.lap_solves = self.lap_fallbacks = 0
        self.lap_events = None
        self.lap_stats = None
        self.lap_winners = np.zeros(32, np.int64)      # wins per racer (JV_RACE_MAX = 28 racers, 5 bits in the statistics)
        self.trans = torch.empty((self.B, self.P, 4, 4), dtype=torch.float32, device=self.dev)
        self.pc_trans = torch.empty((self.B, self.N, 3), dtype=torch.float32, device=self.dev)
        self.G = torch.zeros((self.B, self.N, 3), dtype=torch.float32, device=self.dev)
        # root motion: the articulated cloud before the per-frame rigid motion, and dL/d of it
        self.pc_fk = torch.empty_like(self.pc_trans) if self.root else self.pc_trans
        self.G_fk = torch.empty_like(self.G) if self.root else self.G
        self.losses = {}
        self._pc_src, self._inplace, self._g_pre, self._g_post = None, None, None, None
        self._adam_tab = self._src_idx32 = None
        self._side = None
        # reart_kin_post's inputs that never change: the sample slot of every canonical point, the frames' flow references padded
        # to one length; its buffers (owned here: a captured graph keeps their addresses)
        n = int(self.src_idx.numel())
        self._slot = torch.full((self.N,), -1, dtype=torch.int32, device=self.dev)
        self._slot[self.src_idx] = torch.arange(n, dtype=torch.int32, device=self.dev)
        self._ref_pad = self._flow_pad = self._ref_len = None
        self._nr_max = 0
        if self.refs is not None:
            lens = [int(r.shape[0]) for r, _ in self.refs]
            self._nr_max = max(lens)
            self._ref_pad = torch.zeros((self.B, self._nr_max, 3), dtype=torch.float32, device=self.dev)
            self._flow_pad = torch.zeros((self.B, self._nr_max, 3), dtype=torch.float32, device=self.dev)
            for f, (r, fl) in enumerate(self.refs):
                self._ref_pad[f, :lens[f]] = r
                self._flow_pad[f, :lens[f]] = fl
            if min(lens) != self._nr_max:
                self._ref_len = torch.tensor(lens, dtype=torch.int64, device=self.dev)
        self._matched_buf = torch.zeros((self.B, n, 3), dtype=torch.float32, device=self.dev)
        self._loss3 = torch.zeros((3,), dtype=torch.float32, device=self.dev)
        self._post_ws = torch.empty((max(int(_lib.lib().reart_kin_post_workspace_bytes(self.B, self.N, self._nr_max, 3)), 256),),
                                    dtype=torch.uint8, device=self.dev)

    # ---- pieces -----------------------------------------------------------------------------------------------------
    def _joint_values(self):
        m = self.model
        dist = m.distance_list if hasattr(m, "distance_list") else None
        if self._pris is None:
            return _effective_joint_values(m.theta_list, dist, None)
        # (the mask is built once: _effective_joint_values uploads it on every call, which a captured graph cannot hold)
        return (torch.where(self._pris, torch.full_like(m.theta_list, 1e-6), m.theta_list).detach(),
                torch.where(self._pris, dist, torch.full_like(dist, 1e-6)).detach())

    def forward(self):
        """pc_trans [T-1,N,3] and trans_list of the current parameters (no graph kept)."""
        L = _lib.lib()
        m = self.model
        theta, dist = self._joint_values()
        B, E = theta.shape
        rc = L.reart_fk_forward(_lib.ptr(self.parent), _lib.ptr(self.edge_of), _lib.ptr(self.order), self.P,
                                _lib.ptr(m.axis_list), _lib.ptr(m.moment_list), _lib.ptr(theta), _lib.ptr(dist), B, E,
                                _lib.ptr(self.trans), _lib.stream())
        _lib.check(rc, "reart_fk_forward")
        rc = L.reart_compute_pc_transform(_lib.ptr(self.cano), _lib.ptr(self.trans), _lib.ptr(self.part), self.N, self.P,
                                          B, _lib.ptr(self.pc_fk), _lib.stream())
        _lib.check(rc, "reart_compute_pc_transform")
        if self.root:                                     # networks/model.py:153-158: x R^T + t per frame
            R = self._root_rotation()[0]
            torch.baddbmm(m.root_t.detach()[:, None, :], self.pc_fk, R.transpose(1, 2), out=self.pc_trans)
        return self.pc_trans

    def trans_list(self):
        """The parts' transforms of the last forward, root motion included (what the model's forward returns third)."""
        if not self.root:
            return self.trans
        root = torch.zeros((self.B, 4, 4), dtype=torch.float32, device=self.dev)
        root[:, :3, :3], root[:, :3, 3], root[:, 3, 3] = self._root_rotation()[0], self.model.root_t.detach(), 1.0
        return torch.matmul(root[:, None], self.trans)

    def _root_rotation(self):
        """R [T-1,3,3] of the 6-D parameters (Gram-Schmidt, rows b1, b2, b1 x b2) and what its backward needs."""
        r6 = self.model.root_6d.detach()
        a1, a2 = r6[:, :3], r6[:, 3:]
        n1 = a1.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        b1 = a1 / n1
        s = (b1 * a2).sum(-1, keepdim=True)
        v = a2 - s * b1
        n2 = v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        b2 = v / n2
        return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2), (a2, n1, b1, s, n2, b2)

    def _root_backward(self):
        """self.G = dL/d pc_trans -> gradients of root_6d / root_t, and self.G_fk = dL/d (the cloud before the root motion):
        the hand-derived backward of x R^T + t and of the Gram-Schmidt step (autograd's in OperatorLoop)."""
        m = self.model
        R, (a2, n1, b1, s, n2, b2) = self._root_rotation()
        self.grads[id(m.root_t)].copy_(self.G.sum(dim=1))
        dR = torch.matmul(self.G.transpose(1, 2), self.pc_fk)           # [T-1,3,3]: rows = gradients of b1, b2, b3
        torch.matmul(self.G, R, out=self.G_fk)
        g1, g2, g3 = dR[:, 0], dR[:, 1], dR[:, 2]
        g_b1 = g1 + torch.cross(b2, g3, dim=-1)                         # b3 = b1 x b2
        g_b2 = g2 + torch.cross(g3, b1, dim=-1)
        g_v = (g_b2 - (g_b2 * b2).sum(-1, keepdim=True) * b2) / n2      # b2 = v / |v|
        gvb = (g_v * b1).sum(-1, keepdim=True)
        g_a2 = g_v - gvb * b1                                           # v = a2 - (b1 . a2) b1
        g_b1 = g_b1 - gvb * a2 - s * g_v
        g_a1 = (g_b1 - (g_b1 * b1).sum(-1, keepdim=True) * b1) / n1     # b1 = a1 / |a1|
        self.grads[id(m.root_6d)].copy_(torch.cat((g_a1, g_a2), dim=-1))

    def _backward(self):
        """dL/d pc_trans (self.G) -> gradients of axis / moment / theta (/ distance)."""
        L = _lib.lib()
        m = self.model
        theta, dist = self._joint_values()
        B, E = theta.shape
        g_axis, g_moment, g_theta = self.grads[id(m.axis_list)], self.grads[id(m.moment_list)], self.grads[id(m.theta_list)]
        g_dist = self.grads[id(m.distance_list)] if hasattr(m, "distance_list") else None
        if self.root:
            self._root_backward()
        ws = _lib.workspace(L.reart_fk_backward_workspace_bytes(self.P, B, E), self.dev)
        rc = L.reart_fk_backward(_lib.ptr(self.cano), _lib.ptr(self.part), _lib.ptr(self.G_fk), self.N, _lib.ptr(self.parent),
                                 _lib.ptr(self.edge_of), _lib.ptr(self.order), self.P, _lib.ptr(m.axis_list),
                                 _lib.ptr(m.moment_list), _lib.ptr(theta), _lib.ptr(dist), B, E, _lib.ptr(self.trans),
                                 _lib.ptr(g_axis), _lib.ptr(g_moment), _lib.ptr(g_theta), _lib.ptr(g_dist), _lib.ptr(ws),
                                 ws.numel(), _lib.stream())
        _lib.check(rc, "reart_fk_backward")
        if self._pris is not None:
            g_theta.masked_fill_(self._pris, 0.0)
            g_dist.masked_fill_(~self._pris, 0.0)

    def _adam(self):
        """torch.optim.Adam's step on every parameter tensor (run_robot.py:219-221) in one launch (reart_adam_step_multi)."""
        import ctypes

        L = _lib.lib()
        self.step_count += 1
        grads = []
        for p in self.params:
            g = self.grads[id(p)]
            if self.wd != 0.0:
                g = g + self.wd * p.detach()          # torch.optim.Adam's L2 form
            grads.append(g)
        if self._adam_tab is None or self.wd != 0.0:      # per launch of at most eight tensors: host arrays of pointers, sizes, rates
            self._adam_tab = []
            for a in range(0, len(self.params), 8):
                ps, gs, ms, vs = self.params[a:a + 8], grads[a:a + 8], self.m[a:a + 8], self.v[a:a + 8]
                k = len(ps)
                vp = ctypes.c_void_p * k
                self._adam_tab.append((k, vp(*[p.data_ptr() for p in ps]), vp(*[g.data_ptr() for g in gs]), vp(*[m_.data_ptr() for m_ in ms]),
                                       vp(*[v_.data_ptr() for v_ in vs]), (ctypes.c_int * k)(*[p.numel() for p in ps]),
                                       (ctypes.c_float * k)(*([self.lr] * k))))
        for k, tp, tg, tm, tv, tn, tl in self._adam_tab:
            rc = L.reart_adam_step_multi(k, tp, tg, tm, tv, tn, tl, self.step_count, 0.9, 0.999, 1e-8, _lib.stream())
            _lib.check(rc, "reart_adam_step_multi")

    def _flow_terms(self):
        """lambda_flow * flow_loss and its gradient added to self.G (run_robot.py:194-209)."""
        from .networks.loss import _FlowLoss  # the operator below it is reart_flow_loss

        L = _lib.lib()
        c = self.cano_idx
        comp = torch.cat((self.pc_trans[:c], self.cano[None], self.pc_trans[c:]), dim=0)          # [T,N,3]
        gt = torch.empty((self.B, self.N, 3), dtype=torch.float32, device=self.dev)
        mask = torch.empty((self.B, self.N), dtype=torch.bool, device=self.dev)
        # the T-1 blends are independent and small (16 workgroups each, a chain of gathers): dealt to a few side streams they
        # run side by side -- in a captured graph as parallel branches -- instead of one after the other
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.dev) for _ in range(min(self.SIDE_STREAMS, len(self.refs)))]
        for st in self._side:
            st.wait_stream(cur)
        for f, (r, fl) in enumerate(self.refs):
            with torch.cuda.stream(self._side[f % len(self._side)]):
                ws = _lib.workspace(L.reart_blend_anchor_motion_workspace_bytes(self.N, r.shape[0], 3), self.dev)
                rc = L.reart_blend_anchor_motion(_lib.ptr(comp[f]), _lib.ptr(r), _lib.ptr(fl), self.N, r.shape[0], 3, self.euclid,
                                                 _lib.ptr(gt[f]), _lib.ptr(mask[f]), _lib.ptr(ws), ws.numel(), _lib.stream())
                _lib.check(rc, "reart_blend_anchor_motion")
        for st in self._side:
            cur.wait_stream(st)
        pred = (comp[1:] - comp[:-1]).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=self.dev)
        gp = torch.empty_like(pred)
        ws = _lib.workspace(L.reart_flow_loss_workspace_bytes(), self.dev)
        rc = L.reart_flow_loss(_lib.ptr(gt), _lib.ptr(pred), _lib.ptr(mask), self.B, self.N, int(self.robust), self.smooth,
                               _lib.ptr(loss), _lib.ptr(gp), _lib.ptr(ws), ws.numel(), _lib.stream())
        _lib.check(rc, "reart_flow_loss")
        gp = gp * self.lambda_flow
        # pred = comp[1:] - comp[:-1]: +gp to frame f+1, -gp to frame f; the canonical frame takes no gradient
        gc = torch.zeros_like(comp)
        gc[1:] += gp
        gc[:-1] -= gp
        self.G += torch.cat((gc[:c], gc[c + 1:]), dim=0)
        return loss * self.lambda_flow

    # ---- the iteration ----------------------------------------------------------------------------------------------
    GRAPHS = True       # replay the launches around the solve from two captured graphs (False: every launch eagerly)
    SIDE_STREAMS = 6    # streams the per-frame flow blends of an iteration are dealt to

    def _solve(self, pc_src, behind=None):
        """The assignment refresh (run_robot.py:165-178) -> the solver's [B,4] statistics; the optimum is in lap_state["cols"].
        ``behind``: launches that only READ the optimum (the replay of the post graph), queued behind the re-solve BEFORE the host
        waits for its flags -- the GPU goes straight on while the host wakes up and reads them (75-85 us of every iteration
        otherwise: tools/solve_gaps.py) -- and queued once more in the rare case that the host then changed the columns (a tied
        or an uncertified problem).  Returns True when ``behind`` ran."""
        from .utils import lap

        B, n = pc_src.shape[:2]
        ran = False
        if lap.InPlaceResolve.usable(self.lap_state, B, n):
            if self._inplace is None:
                self._inplace = lap.InPlaceResolve(B, n, self.dev)
            self._inplace.begin(pc_src, self.tgt_pts, self.lap_state, stats=True)
            if self.lap_events is not None:
                self.lap_events[-1][1].record()
            if behind is not None:
                behind()
                ran = True
            fb, raw, changed = self._inplace.finish()
            if changed and ran:
                behind()
            st = raw.copy()
            self.lap_state["commit_conflicts"] = (st[:, 1] >> 16) & 0xffff
            st[:, 1] &= 0xffff
            self.lap_state["backward_rounds"] = (st[:, 0] >> 21) & 0x3ff
            st[:, 0] &= 0x1fffff
            self.lap_state["winner"] = (st[:, 0] >> 16) & 31
        else:       # the first solve (cold), sizes outside the chain forms: the general entry (its columns are a new tensor)
            _, fb, st = lap.linear_sum_assignment_points(pc_src.contiguous(), self.tgt_pts, self.lap_state, return_stats="full", device_cols=True)
            self._g_post = None                     # a graph that reads the columns must see the new tensor
            if self.lap_events is not None:
                self.lap_events[-1][1].record()
        self.lap_solves += 1
        self.lap_fallbacks += fb
        self.lap_stats = st
        # sequential steps of this solve per problem (slowest, mean) and the slowest problem's search steps alone -- the latency
        # roofline of bench.py multiplies them by the measured floor of one step
        seq = st[:, 2].astype(np.int64)
        search_only = int(seq.max())
        if self.lap_state.get("resolve_form", "jv") == "jv":           # one row at a time: the row reduction's steps are sequential too
            seq = seq + (st[:, 3].astype(np.int64) >> 8)
        elif self.lap_state.get("backward_rounds") is not None:        # the backward growth before the searches: the same kind of step
            seq = seq + np.asarray(self.lap_state["backward_rounds"], dtype=np.int64)
        self.lap_steps_log.append((int(seq.max()), float(seq.mean()), search_only))
        self.lap_winners += np.bincount((st[:, 0] >> 16) & 31, minlength=32)[:32]                  # raced re-solves: who finished first
        return ran

    def _pre(self):
        """Forward of the current parameters and the sampled source points (the solver's input)."""
        self.forward()
        if self._src_idx32 is None:
            self._src_idx32 = self.src_idx.int().contiguous()
        rc = _lib.lib().reart_gather_points(_lib.ptr(self.pc_trans), _lib.ptr(self._src_idx32), self.B, self.N, int(self._src_idx32.numel()),
                                            _lib.ptr(self._pc_src), _lib.stream())                 # pc_trans[:, src_idx], one launch
        _lib.check(rc, "reart_gather_points")

    FUSED_POST = True   # reart_kin_post (nine launches); False: the same values as tensor expressions (_post_expressions)

    def _post(self):
        """Everything between the solve and Adam: matched targets, assignment (+ flow) loss, dL/d pc_trans, the FK backward."""
        if not self.FUSED_POST or self.lap_state["cols"].dtype != torch.int32 or (self.refs is not None and min(int(r.shape[0]) for r, _ in self.refs) < 3):
            return self._post_expressions()
        L = _lib.lib()
        n = int(self.src_idx.numel())
        rc = L.reart_kin_post(_lib.ptr(self.pc_trans), _lib.ptr(self.cano), self.B, self.N, self.cano_idx, _lib.ptr(self._pc_src),
                              _lib.ptr(self.tgt_pts), _lib.ptr(self.lap_state["cols"]), _lib.ptr(self._slot), n, self.lambda_assign,
                              _lib.ptr(self._ref_pad), _lib.ptr(self._flow_pad), _lib.ptr(self._ref_len), self._nr_max, 3, self.euclid,
                              self.lambda_flow, int(self.robust), self.smooth, _lib.ptr(self.G), _lib.ptr(self._matched_buf),
                              _lib.ptr(self._loss3), _lib.ptr(self._post_ws), self._post_ws.numel(), _lib.stream())
        _lib.check(rc, "reart_kin_post")
        self.matched = self._matched_buf
        losses = {"opt assignment loss": self._loss3[0]}
        if self.refs is not None:
            losses["flow Loss"] = self._loss3[1]
        losses["total Loss"] = self._loss3[2]
        self._backward()
        self.losses = losses

    def _post_expressions(self):
        """_post as the reference's tensor expressions (run_robot.py:177-209) over the per-frame operators: the form reart_kin_post
        replaced, kept as its check (tests/test_kinematic_engine_gpu.py)."""
        cols = self.lap_state["cols"].long()
        self.matched = self.tgt_pts.gather(1, cols[..., None].expand(-1, -1, 3))
        diff = self._pc_src - self.matched
        ass = self.lambda_assign * (diff * diff).sum()                                            # run_robot.py:181-184
        self.G.zero_()
        self.G[:, self.src_idx] = (2.0 * self.lambda_assign) * diff
        losses = {"opt assignment loss": ass}
        total = ass
        if self.refs is not None:
            fl = self._flow_terms()
            losses["flow Loss"] = fl
            total = total + fl
        losses["total Loss"] = total
        self._backward()
        self.losses = losses

    def _post_chamfer(self):
        """The Chamfer branch (run_robot.py:187-190: recon_loss = sum of the bidirectional per-point Chamfer distance,
        networks/loss.py:24-29) without autograd: both searches, then the gradient of each direction's distances w.r.t. the
        predicted cloud (utils/chamfer.py:195-209 with grad_dists = 1), flow terms, FK backward."""
        from . import chamferdist_C as _C

        X, Y = self.pc_trans, self.pc_list
        i1, d1 = _C.knn_points_idx(X, Y, None, None, 1)
        i2, d2 = _C.knn_points_idx(Y, X, None, None, 1)
        rec = d1.sum() + d2.sum()
        ones = torch.ones_like(d1)
        g1, _ = _C.knn_points_backward(X, Y, None, None, i1, ones)
        _, g2 = _C.knn_points_backward(Y, X, None, None, i2, ones)
        self.G.copy_(g1)
        self.G += g2
        losses = {"recon Loss": rec}
        total = rec
        if self.refs is not None:
            fl = self._flow_terms()
            losses["flow Loss"] = fl
            total = total + fl
        losses["total Loss"] = total
        self._backward()
        self.losses = losses

    @staticmethod
    def _graph(fn):
        fn()                                        # warm-up outside the capture (lazy loads, workspace growth)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        return g

    @torch.no_grad()
    def iteration(self, i):
        """Iteration i of run_robot.py:154-221 in the assignment-loss branch; returns the loss dictionary (device tensors).
        From the third iteration on the launches before and after the solve replay from two captured graphs (the solve itself
        and Adam, whose step count is a host value, stay eager): same launches, same results, a tenth of the host work."""
        if not self.use_assign or i < self.assign_iter:                   # run_robot.py:187-190
            self.forward()
            self._post_chamfer()
            self._adam()
            return self.losses
        if self._pc_src is None:
            self._pc_src = torch.empty((self.B, self.src_idx.numel(), 3), dtype=torch.float32, device=self.dev)
        if self._g_pre is not None:
            self._g_pre.replay()
        else:
            self._pre()
            if self.GRAPHS and self.lap_solves >= 2:
                self._g_pre = self._graph(self._pre)                      # (the warm-up inside recomputes the same forward)
        posted = False
        if self.lap_state.get("cols") is None or i % self.assign_gap == 0:                       # run_robot.py:165-178
            if self.lap_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                self.lap_events.append(ev)              # (the end is recorded right behind the re-solve's launches)
            posted = self._solve(self._pc_src, behind=self._g_post.replay if self._g_post is not None else None)
        if posted:
            pass                                        # (the post graph went in behind the re-solve)
        elif self._g_post is not None:
            self._g_post.replay()
        elif self.GRAPHS and self.lap_solves >= 2 and self._inplace is not None:
            self._g_post = self._graph(self._post)                        # (warm-up + capture; the replay below does the work once more:
            self._g_post.replay()                                         #  _post only writes buffers it fully overwrites)
        else:
            self._post()
        self._adam()
        return self.losses
