"""Fused relaxation engine: the reference's optimisation loop body (run_robot.py:154-221,
Chamfer [+ flow] branch) as five HIP kernel launches per iteration, optionally replayed from a
captured graph.  ``RelaxEngine`` owns the torch tensors (device memory) and hands raw pointers
to ``reart_relax_prepare`` / ``reart_relax_step`` (include/reart_hip.h)."""
import ctypes
import os

import torch

from . import _lib

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class RelaxConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("N", "P", "B", "H", "cano_idx", "use_flow", "robust", "euclidean", "flow_k",
                                     "M_max", "M_total", "n_iter", "ring")] + \
               [(n, c_float) for n in ("lambda_flow", "smooth_weight", "trans_lr", "seg_lr", "beta1", "beta2", "eps",
                                       "start_tau", "end_tau", "fixed_tau")] + [("seed", ctypes.c_uint64),
                                                                                 ("use_grid", c_int), ("use_boxes", c_int),
                                                                                 ("use_assign", c_int), ("lambda_assign", c_float),
                                                                                 ("weight_decay", c_float)] + \
               [(n, c_int) for n in ("search_mode", "tune_slices", "tune_slices_flow", "tune_sparse", "tune_fwd_pts",
                                     "tune_bwd_pts", "tune_reorder", "tune_cloud", "tune_xcd", "profile", "tune_share")]


def tuning_from_env(env=None):
    """Experiment switches of the library, read ONCE per engine on the host (the library itself never looks at the
    environment): REART_SEARCH=brute, REART_PRUNE_SPLIT / REART_PRUNE_SPLIT3 (waves per search workgroup, 1..4),
    REART_SPARSE (0 = dense scans only, 1..16), REART_FWD_PTS (64|32), REART_BWD_PTS (64|32|16), REART_REORDER=0, REART_CLOUD=1|2|4|8 (cloud-resident search: targets from an LDS copy, that many box slices per query group), REART_XCD=1 (one run of frames per XCD)."""
    env = os.environ if env is None else env
    geti = lambda k: int(env[k]) if env.get(k, "") != "" else None
    t = {}
    if env.get("REART_SEARCH") == "brute":
        t["search_mode"] = 1
    for key, field in (("REART_PRUNE_SPLIT", "tune_slices"), ("REART_PRUNE_SPLIT3", "tune_slices_flow"),
                       ("REART_FWD_PTS", "tune_fwd_pts"), ("REART_BWD_PTS", "tune_bwd_pts")):
        if geti(key) is not None:
            t[field] = geti(key)
    if geti("REART_SPARSE") is not None:
        t["tune_sparse"] = -1 if geti("REART_SPARSE") <= 0 else geti("REART_SPARSE")
    if env.get("REART_REORDER") == "0":
        t["tune_reorder"] = -1
    if geti("REART_CLOUD") is not None:
        t["tune_cloud"] = max(0, geti("REART_CLOUD"))
    if env.get("REART_XCD") == "1":
        t["tune_xcd"] = 1
    if env.get("REART_SHARE") in ("0", "1"):        # 0: own seeds only; 1: neighbour seeds, every wave all candidates
        t["tune_share"] = -1 if env["REART_SHARE"] == "0" else 1
    return t


class RelaxBuffers(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("cano", "pc_list", "ref_loc", "ref_flow", "ref_off", "gumbel", "W1", "b1", "W2",
                                        "p6d", "pt", "adam_m", "adam_v", "iter", "tau", "losses", "pc_trans",
                                        "seg_part", "trans_list", "aux_stream", "ev_fork", "ev_join", "assign_map")]


def morton_order(points):
    """Permutation that sorts a cloud [N,3] along a 30-bit Morton (Z-order) curve.  Setup-time
    plumbing: waves of 64 consecutive points become spatially compact, which is what makes the
    bounding-box block-skip test of the K-NN kernels effective.  Deterministic (stable sort)."""
    p = points.detach().float()
    lo, hi = p.min(dim=0).values, p.max(dim=0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-20).max() * 1023.0).clamp(0, 1023).long()

    def spread(v):  # 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.sort(code, stable=True).indices


def kd_order(points, leaf=16):
    """Permutation that stores a cloud [N,3] in the leaf order of a balanced k-d tree (median split
    along the longest axis, leaves of ``leaf`` = NN_BOX points, left halves rounded up to whole leaves).
    Every 16 consecutive points then form a tight box and every 64 a compact group: on the loop's
    clouds the box-pruned searches look at about half the boxes a Morton order needs.  Setup-time
    plumbing on the host (N log N, numpy); deterministic (stable sorts)."""
    import numpy as np

    p = points.detach().float().cpu().numpy()
    out, stack = [], [np.arange(p.shape[0])]
    while stack:
        ids = stack.pop()
        if ids.shape[0] <= leaf:
            out.append(ids)
            continue
        pts = p[ids]
        ax = int(np.argmax(pts.max(0) - pts.min(0)))
        o = ids[np.argsort(pts[:, ax], kind="stable")]
        half = (o.shape[0] // 2 + leaf - 1) // leaf * leaf
        stack.append(o[half:])   # popped after the left half: leaves come out left to right
        stack.append(o[:half])
    return torch.from_numpy(np.concatenate(out)).to(points.device)


_L = None


def _lib_fns():
    global _L
    if _L is None:
        L = _lib.lib()
        L.reart_relax_workspace_bytes.restype = ctypes.c_size_t
        L.reart_relax_workspace_bytes.argtypes = [ctypes.POINTER(RelaxConfig)]
        for fn in (L.reart_relax_prepare, L.reart_relax_step, L.reart_relax_forward):
            fn.restype = c_int
            fn.argtypes = [ctypes.POINTER(RelaxConfig), ctypes.POINTER(RelaxBuffers), c_void_p, ctypes.c_size_t, c_void_p]
        L.reart_relax_step_timed.restype = c_int
        L.reart_relax_step_timed.argtypes = [ctypes.POINTER(RelaxConfig), ctypes.POINTER(RelaxBuffers), c_void_p,
                                             ctypes.c_size_t, c_void_p, ctypes.POINTER(c_float)]
        L.reart_relax_step_batch.restype = c_int
        L.reart_relax_step_batch.argtypes = [ctypes.POINTER(RelaxConfig), ctypes.POINTER(RelaxBuffers), ctypes.POINTER(c_void_p),
                                             ctypes.c_size_t, c_int, c_void_p]
        L.reart_relax_profile.restype = c_int
        L.reart_relax_profile.argtypes = [ctypes.POINTER(RelaxConfig), c_void_p, ctypes.c_size_t, c_void_p,
                                          ctypes.POINTER(ctypes.c_double), c_int]
        _L = L
    return _L


def gumbel_noise(seed, iteration, N, P, device):
    """[N,P] Gumbel samples exactly as the fused step draws them in its forward kernel for iteration ``iteration`` of an
    engine seeded with ``seed`` (``reart_gumbel_noise``; the stand-in for the draw inside F.gumbel_softmax,
    networks/model.py:44).  Row n belongs to the engine's n-th STORED point (k-d leaf order when ``spatial_sort``)."""
    out = torch.empty((N, P), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        rc = _lib.lib().reart_gumbel_noise(int(seed), int(iteration), N, P, _lib.ptr(out), _lib.stream())
    _lib.check(rc, "reart_gumbel_noise")
    return out


class RelaxEngine:
    """State of one optimisation instance on one GPU.

    cano_pc [N,3], pc_list [T-1,N,3]; ``model`` is a ``reart_amd.networks.model.BaseModel`` whose
    parameters are optimised IN PLACE; ``pc_ref_list`` / ``flow_ref_list`` are the ragged flow
    references of run_robot.py:81-84 (lists of [M_i,3] tensors) or None for Chamfer only.
    Hyper-parameters carry the reference's flag names and defaults (run_robot.py:362-420).
    """

    def __init__(self, cano_pc, pc_list, model, cano_idx, pc_ref_list=None, flow_ref_list=None, n_iter=15000,
                 start_tau=5.0, end_tau=1.0, trans_lr=1e-2, seg_lr=1e-3, lambda_flow=1.0, use_robust_loss=False,
                 smooth_weight=1e-2, fixed_tau=0.0, seed=2, ring=1024, knn_squared=False, start_iter=0, use_grid=False,
                 overlap_flow=True, spatial_sort=True, weight_decay=0.0, profile=False, tuning=None):
        _lib.require_gpu(cano_pc, pc_list)
        dev = cano_pc.device
        self.device, self.model = dev, model
        cano_pc, pc_list = cano_pc.float(), pc_list.float()
        # Internal storage order: every cloud in the leaf order of its own k-d tree (results are returned
        # in the caller's order; the optimisation problem is invariant to point order).
        order = {"kd": kd_order, "morton": morton_order}[os.environ.get("REART_ORDER", "kd")]
        self._perm = order(cano_pc) if spatial_sort else None
        self._perm_frames = None
        if spatial_sort:
            cano_pc = cano_pc[self._perm]
            self._perm_frames = torch.stack([order(f) for f in pc_list])
            pc_list = torch.stack([f[o] for f, o in zip(pc_list, self._perm_frames)])
            if pc_ref_list is not None:
                orders = [order(r.reshape(-1, 3)) for r in pc_ref_list]
                pc_ref_list = [r.reshape(-1, 3)[o] for r, o in zip(pc_ref_list, orders)]
                flow_ref_list = [f.reshape(-1, 3)[o] for f, o in zip(flow_ref_list, orders)]
            self._inv = torch.empty_like(self._perm)
            self._inv[self._perm] = torch.arange(self._perm.numel(), device=dev)
        self.cano = cano_pc.contiguous()
        self.pc_list = pc_list.contiguous()
        B, N, _ = self.pc_list.shape
        c1, c2 = model.seg_head.model[0], model.seg_head.model[2]
        H, P = c1.weight.shape[0], c2.weight.shape[0]
        self.params = [c1.weight, c1.bias, c2.weight, model.proposal_6d, model.proposal_t]
        for p in self.params:
            assert p.is_cuda and p.is_contiguous() and p.dtype == torch.float32
        nparams = sum(p.numel() for p in self.params)
        self.adam_m = torch.zeros(nparams, device=dev)
        self.adam_v = torch.zeros(nparams, device=dev)
        self.iter = torch.full((1,), int(start_iter), dtype=torch.int64, device=dev)
        self.tau = torch.zeros(1, device=dev)
        self.ring = ring
        self.losses = torch.zeros((ring, 4), device=dev)
        self._pc_trans = torch.empty((B, N, 3), device=dev)
        self._assign_map = torch.full((B, N), -1, dtype=torch.int32, device=dev)
        self._seg_part = torch.empty((N,), dtype=torch.int64, device=dev)
        self.trans_list = torch.empty((B, P, 4, 4), device=dev)
        self.gumbel = None
        use_flow = pc_ref_list is not None
        if use_flow:
            assert len(pc_ref_list) == B and len(flow_ref_list) == B
            lens = [int(r.shape[0]) for r in pc_ref_list]
            self.ref_loc = torch.cat([r.reshape(-1, 3) for r in pc_ref_list]).contiguous().float().to(dev)
            self.ref_flow = torch.cat([r.reshape(-1, 3) for r in flow_ref_list]).contiguous().float().to(dev)
            off = [0]
            for m in lens:
                off.append(off[-1] + m)
            self.ref_off = torch.tensor(off, dtype=torch.int32, device=dev)
        else:
            lens, self.ref_loc, self.ref_flow, self.ref_off = [0], None, None, None
        self.cfg = RelaxConfig(N=N, P=P, B=B, H=H, cano_idx=cano_idx, use_flow=int(use_flow),
                               robust=int(bool(use_robust_loss)), euclidean=0 if knn_squared else 1, flow_k=3,
                               M_max=max(lens), M_total=sum(lens), n_iter=n_iter, ring=ring, lambda_flow=lambda_flow,
                               smooth_weight=smooth_weight, trans_lr=trans_lr, seg_lr=seg_lr, beta1=0.9, beta2=0.999,
                               eps=1e-8, start_tau=start_tau, end_tau=end_tau, fixed_tau=fixed_tau, seed=seed,
                               use_grid=int(bool(use_grid)), use_boxes=int(bool(spatial_sort)),
                               weight_decay=float(weight_decay), profile=int(bool(profile)))
        for k, v in (tuning_from_env() if tuning is None else tuning).items():
            setattr(self.cfg, k, int(v))
        L = _lib_fns()
        nbytes = L.reart_relax_workspace_bytes(ctypes.byref(self.cfg))
        if nbytes == 0:
            raise _lib.ReartHipError("reart_relax_workspace_bytes: unsupported configuration")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._graph = None
        self._bufs = None
        self.graph_replays = self.eager_steps = 0
        # fork/join resources for running the flow branch beside the Chamfer search (caller-owned)
        self._aux = None
        if use_flow and overlap_flow:
            self._aux = torch.cuda.Stream(device=dev)
            self._ev = (torch.cuda.Event(), torch.cuda.Event())
            for ev in self._ev:
                ev.record()  # materialise the hipEvent_t handles
        self._refresh_buffers()
        rc = L.reart_relax_prepare(ctypes.byref(self.cfg), ctypes.byref(self._bufs), _lib.ptr(self.workspace),
                                   self.workspace.numel(), _lib.stream())
        _lib.check(rc, "reart_relax_prepare")

    def _refresh_buffers(self):
        p = _lib.ptr
        v = lambda t: None if t is None else t.data_ptr()
        W1, b1, W2, p6d, pt = self.params
        self._bufs = RelaxBuffers(cano=v(self.cano), pc_list=v(self.pc_list), ref_loc=v(self.ref_loc),
                                  ref_flow=v(self.ref_flow), ref_off=v(self.ref_off), gumbel=v(self.gumbel),
                                  W1=v(W1), b1=v(b1), W2=v(W2), p6d=v(p6d), pt=v(pt), adam_m=v(self.adam_m),
                                  adam_v=v(self.adam_v), iter=v(self.iter), tau=v(self.tau), losses=v(self.losses),
                                  pc_trans=v(self._pc_trans), seg_part=v(self._seg_part), trans_list=v(self.trans_list),
                                  aux_stream=None if self._aux is None else self._aux.cuda_stream,
                                  ev_fork=None if self._aux is None else self._ev[0].cuda_event,
                                  ev_join=None if self._aux is None else self._ev[1].cuda_event,
                                  assign_map=v(self._assign_map))

    @property
    def pc_trans(self):
        """[T-1,N,3] forward output of the last iteration, in the caller's point order."""
        return self._pc_trans if self._perm is None else self._pc_trans[:, self._inv]

    @property
    def seg_part(self):
        """[N] arg-max part of the noise-free logits (last iteration), in the caller's point order."""
        return self._seg_part if self._perm is None else self._seg_part[self._inv]

    def caller_clouds(self):
        """(cano_pc [N,3], pc_list [T-1,N,3]) in the CALLER's point order (the engine stores every cloud in the leaf
        order of its own k-d tree).  Anything that depends on point order -- per-part FPS, tie rules of the structure
        tail and of the linear assignment -- must see these, not ``self.cano`` / ``self.pc_list``."""
        if self._perm is None:
            return self.cano, self.pc_list
        B, N = self._perm_frames.shape
        inv_f = torch.empty_like(self._perm_frames)
        inv_f.scatter_(1, self._perm_frames, torch.arange(N, device=self.device).expand(B, N))
        return self.cano[self._inv], torch.gather(self.pc_list, 1, inv_f[..., None].expand(-1, -1, 3))

    def set_gumbel(self, noise):
        """Inject the Gumbel noise [N,P] used by every following step (tests); None = in-kernel Philox."""
        assert self._graph is None, "noise injection is an eager-mode (test) feature"
        if noise is not None and self._perm is not None:
            noise = noise[self._perm]
        self.gumbel = None if noise is None else noise.contiguous().float()
        self._refresh_buffers()

    # ---- assignment loss (run_robot.py:164-187): the caller refreshes the pairs every assign_gap iterations
    def peek_forward(self):
        """Forward of the CURRENT iteration only (the temperature and Gumbel noise the next ``step`` will use);
        afterwards ``pc_trans`` / ``seg_part`` hold that iteration's output.  Changes nothing else."""
        rc = _lib_fns().reart_relax_forward(ctypes.byref(self.cfg), ctypes.byref(self._bufs), _lib.ptr(self.workspace),
                                            self.workspace.numel(), _lib.stream())
        _lib.check(rc, "reart_relax_forward")

    def set_assignment(self, src_idx, tgt_idx, lambda_assign):
        """Switch to the assignment loss ``lambda_assign * sum |pc_trans[b, src_idx[r]] - pc_list[b, tgt_idx[b, r]]|^2``.
        src_idx [n] (indices into the canonical cloud), tgt_idx [B,n] (indices into pc_list[b]), both in the
        caller's point order.  A captured graph is dropped when the mode changes (capture again)."""
        B, N = self._assign_map.shape
        src = src_idx.to(self.device).long().reshape(-1)
        tgt = tgt_idx.to(self.device).long().reshape(B, -1)
        if self._perm is not None:   # caller order -> internal storage order
            src = self._inv[src]
            if getattr(self, "_inv_frames", None) is None:      # fixed for the engine's life: built on first use
                inv_f = torch.empty_like(self._perm_frames)
                inv_f.scatter_(1, self._perm_frames, torch.arange(N, device=self.device).expand(B, N))
                self._inv_frames = inv_f
            tgt = self._inv_frames.gather(1, tgt)
        self._assign_map.fill_(-1)
        self._assign_map[:, src] = tgt.to(torch.int32)
        lam32 = ctypes.c_float(lambda_assign).value       # the config holds fp32: compare what it would hold (0.3 != fp32(0.3))
        if not self.cfg.use_assign or self.cfg.lambda_assign != lam32:
            self.cfg.use_assign, self.cfg.lambda_assign = 1, float(lambda_assign)
            self._graph = None

    def clear_assignment(self):
        if self.cfg.use_assign:
            self.cfg.use_assign = 0
            self._graph = None

    def _enqueue(self):
        rc = _lib_fns().reart_relax_step(ctypes.byref(self.cfg), ctypes.byref(self._bufs), _lib.ptr(self.workspace),
                                         self.workspace.numel(), _lib.stream())
        _lib.check(rc, "reart_relax_step")

    def capture(self, steps_per_graph=1):
        """Capture ``steps_per_graph`` iterations into one graph; ``step()`` then replays it."""
        self._enqueue()  # warm-up outside capture (lazy module load)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (an RCCL watchdog) may touch the runtime
            for _ in range(steps_per_graph):
                self._enqueue()
        self._graph, self._steps_per_graph = g, steps_per_graph
        return 1  # iterations consumed by the warm-up

    def step(self, n=1):
        """Enqueue n iterations (asynchronous; no host sync).  Counts what it did in ``graph_replays`` /
        ``eager_steps`` (the remainder of n over the captured graph's length runs eagerly: same launches, same results)."""
        if self._graph is not None:
            for _ in range(n // self._steps_per_graph):
                self._graph.replay()
            self.graph_replays += n // self._steps_per_graph
            n %= self._steps_per_graph
        for _ in range(n):
            self._enqueue()
        self.eager_steps += n

    PHASES = ("forward", "flow_knn3", "flow_blend", "chamfer_nn", "chamfer_grad", "backward", "adam", "bookkeep")

    def step_timed(self, n=1):
        """Run n eager iterations with hipEvents between phases (on the launch stream); returns
        {phase: mean milliseconds per iteration}.  Synchronises; for measurement only."""
        acc = (c_float * len(self.PHASES))()
        for _ in range(n):
            rc = _lib_fns().reart_relax_step_timed(ctypes.byref(self.cfg), ctypes.byref(self._bufs),
                                                   _lib.ptr(self.workspace), self.workspace.numel(), _lib.stream(), acc)
            _lib.check(rc, "reart_relax_step_timed")
        return {k: acc[i] / n for i, k in enumerate(self.PHASES)}

    def search_profile(self, reset=False):
        """``profile=True`` engines: what the search launches did since the last reset, measured on the device in every
        iteration (eager or graph replay) -> dict(launches, seconds, pairs, clock_hz); seconds = sum over launches of
        (last workgroup end - first workgroup start).  Synchronises."""
        out = (ctypes.c_double * 5)()
        rc = _lib_fns().reart_relax_profile(ctypes.byref(self.cfg), _lib.ptr(self.workspace), self.workspace.numel(),
                                            _lib.stream(), out, int(bool(reset)))
        _lib.check(rc, "reart_relax_profile")
        return {"launches": int(out[0]), "seconds": float(out[1]), "pairs": int(out[2]), "clock_hz": float(out[3]),
                "workgroup_seconds": float(out[4])}

    def loss_log(self):
        """(iterations done, tensor [min(iter, ring), 4]: recon, lambda*flow, total, tau); syncs."""
        it = int(self.iter.item())
        rows = self.losses[: min(it, self.ring)] if it <= self.ring else torch.roll(self.losses, -(it % self.ring), 0)
        return it, rows.clone()

    def last_losses(self):
        it = int(self.iter.item())
        return self.losses[(it - 1) % self.ring].clone()


class RelaxBatch:
    """Up to ``MAX`` engines of ONE shape stepping together: every kernel of the iteration is launched once for all of them
    (``reart_relax_step_batch``), each engine computing exactly what its own ``step()`` would.  The sweep over canonical
    frames (README.md:60) runs its instances this way -- one instance leaves most of the 256 compute units idle.
    The engines keep their own state; read results from them as usual."""

    MAX = 6

    def __init__(self, engines):
        engines = list(engines)
        if not 1 <= len(engines) <= self.MAX:
            raise ValueError(f"RelaxBatch takes 1..{self.MAX} engines")
        nb = {e.workspace.numel() for e in engines}
        if len(nb) != 1:
            raise ValueError("RelaxBatch: the engines must share one shape")
        # the shared launches take their grid and their code path from instance 0: every engine must agree on everything
        # that decides either (two shapes can need the same workspace size)
        for f in self.SAME:
            vals = {getattr(e.cfg, f) for e in engines}
            if len(vals) != 1:
                raise ValueError(f"RelaxBatch: the engines differ in cfg.{f} ({sorted(vals)})")
        if len({e.device for e in engines}) != 1:
            raise ValueError("RelaxBatch: the engines live on different devices")
        self.engines = engines
        self._nbytes = nb.pop()
        self._graph = None
        self.graph_replays = self.eager_steps = 0
        self._refresh()

    # shape and switch fields of reart_relax_config that every engine of a batch must share
    SAME = ("N", "P", "B", "H", "M_max", "use_flow", "robust", "euclidean", "flow_k", "use_grid", "use_boxes", "use_assign",
            "search_mode", "tune_slices", "tune_slices_flow", "tune_sparse", "tune_fwd_pts", "tune_bwd_pts", "tune_reorder",
            "tune_cloud", "tune_xcd", "tune_share")

    def _refresh(self):
        """The argument blocks are VALUES: re-read them from the engines (an engine may have been given new noise, new
        buffers or another loss since the batch was built).  A captured graph keeps the values it was captured with, like
        an engine's own graph does."""
        K = len(self.engines)
        self._cfgs = (RelaxConfig * K)(*[e.cfg for e in self.engines])
        self._bufs = (RelaxBuffers * K)(*[e._bufs for e in self.engines])
        self._ws = (c_void_p * K)(*[e.workspace.data_ptr() for e in self.engines])

    def _enqueue(self):
        if len({(e.cfg.use_assign, e.cfg.use_flow) for e in self.engines}) != 1:
            raise RuntimeError("RelaxBatch: the engines must be in the same loss mode (all Chamfer or all assignment loss)")
        self._refresh()
        rc = _lib_fns().reart_relax_step_batch(self._cfgs, self._bufs, self._ws, self._nbytes, len(self.engines), _lib.stream())
        _lib.check(rc, "reart_relax_step_batch")

    def capture(self, steps_per_graph=1):
        """Capture ``steps_per_graph`` iterations of all engines into one graph; ``step()`` then replays it."""
        self._enqueue()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (an RCCL watchdog) may touch the runtime
            for _ in range(steps_per_graph):
                self._enqueue()
        self._graph, self._steps_per_graph = g, steps_per_graph
        self._graph_mode = self._mode()
        return 1

    def _mode(self):
        """What a captured graph has baked in besides the buffers: the loss mode of every engine."""
        return tuple((e.cfg.use_assign, e.cfg.use_flow, e.cfg.lambda_assign) for e in self.engines)

    def step(self, n=1):
        """n iterations of every engine (asynchronous)."""
        if self._graph is not None and self._graph_mode != self._mode():
            self._graph = None          # an engine changed its loss since the capture (set_assignment): that graph is another iteration
        if self._graph is not None:
            for _ in range(n // self._steps_per_graph):
                self._graph.replay()
            self.graph_replays += n // self._steps_per_graph
            n %= self._steps_per_graph
        for _ in range(n):
            self._enqueue()
        self.eager_steps += n
