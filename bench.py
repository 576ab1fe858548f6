#!/usr/bin/env python3
"""Headline benchmark: relaxation-loop iterations/sec on the synthetic T=20 x N=4096 sequence
(BASELINE.json configs[1]: Chamfer + flow loss), one optimisation instance per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of the reference loop body (run_robot.py:154-221): model
forward with fresh Gumbel noise, bidirectional Chamfer, 19 k=3 flow blends + flow loss, backward,
Adam -- nothing skipped, inputs resident in HBM, replayed from a captured graph.  Multi-GPU is the
reference's natural sharding: independent (sequence, cano_idx) instances, one per rank, no
data-path collective; only the final energies are gathered (RCCL all_gather).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _launch_module():
    """reart_amd/launch.py loaded by path: the parent of a self-launched job imports neither torch nor the package."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_reart_launch", os.path.join(ROOT, "reart_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__" and _requested_gpus(sys.argv[1:]) > 1 and "RANK" not in os.environ:
    # `python bench.py --gpus N` with no launcher around it: start the N ranks (one per GPU, RCCL) BEFORE anything in this
    # process touches the GPU, relay rank 0's JSON line, fail loudly if the ranks fail or the line is not an N-GPU line
    sys.exit(_launch_module().self_launch(os.path.abspath(__file__), sys.argv[1:], _requested_gpus(sys.argv[1:])))

import numpy as np  # noqa: E402
import torch  # noqa: E402

P_BENCH = 20             # parts of the bench model (BaseModel(num_parts=20))
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector peak (counts FMA as 2)


def _sig(x, n=5, top=True):
    """Floats below the top level rounded to n significant digits (the line is read by people and stored in a 5 KB record)."""
    if isinstance(x, dict):
        return {k: (v if top and k in ("value", "ms_per_step", "per_gpu") else _sig(v, n, False)) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n, False) for v in x]
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")) and x != 0.0:
        return float(f"{x:.{n}g}")
    return x


def _line(out):
    return json.dumps(_sig(out), separators=(",", ":"))


def _pcts(ms):
    """p50 / p95 / max of per-solve milliseconds (the mean sits next to them as kernel_ms)."""
    if not len(ms):
        return {}
    a = np.asarray(ms, dtype=np.float64)
    return {"solve_ms_p50": round(float(np.percentile(a, 50)), 4), "solve_ms_p95": round(float(np.percentile(a, 95)), 4),
            "solve_ms_max": round(float(a.max()), 4)}


def _ties_of(lap_state):
    """[problems flagged by reart_lap_ties, problems whose assignment the canonical choice changed] of a loop, or None when the
    loop ran without the tie check (--no-deterministic)."""
    tb = (lap_state or {}).get("tie_breaker")
    return None if tb is None else [int(tb.flagged), int(tb.changed)]


def step_chain_floor(dev, N, B, P=20, H=128):
    """The headline iteration's launch chain with its arithmetic removed (reart_relax_step_floor): five dependent launches with
    the grids, block sizes and LDS footprints of the step's kernels at this configuration (rocprofv3 kernel trace) and the
    dependent global round trips / workgroup barriers of one workgroup's chain in each (read off the kernels: forward 1 / 6,
    search 11 / 2, consumers 3 / 3, backward 1 / 6, finalize 2 / 2), replayed from a graph like the step
    itself -> (microseconds per iteration of the five launches, of the four small kernels' launches alone).  What is above it in
    `ms_per_step` is the kernels' arithmetic and whatever of their chains the counts miss."""
    import ctypes

    from reart_amd import _lib as L_
    nfin = ((4 * (P * H + 4 * H) + 63) // 64 * 64 + 64 * B * P + 255) // 256
    shape = [N // 32, 320, 53760, 1, 6,
             3 * B * (N // 64), 192, 17664, 11, 2,
             B * (N // 512) + B * (N // 1024) + 4, 1024, 33792, 3, 3,
             N // 32, 640, 65536, 1, 6,
             nfin, 384, 512, 2, 2]
    ws = torch.empty(65536 + 256, dtype=torch.uint8, device=dev)
    per, reps, out = 50, 20, []
    for sh in (shape, shape[:5] + shape[10:]):          # all five launches | the four small kernels alone (without the search)
        arr = (ctypes.c_int * len(sh))(*sh)
        run = lambda: L_.check(L_.lib().reart_relax_step_floor(arr, len(sh) // 5, per, L_.ptr(ws), ws.numel(), L_.stream()), "reart_relax_step_floor")
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            run()
        for _ in range(3):
            g.replay()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            g.replay()
        ev1.record()
        torch.cuda.synchronize()
        out.append(1e3 * ev0.elapsed_time(ev1) / (reps * per))
    return out[0], out[1]


def build_instance(dev, T, N, cano_idx, seed, use_flow=True, n_iter=15000, use_grid=False, overlap=True, profile=False):
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.synthetic import make_sequence, split_canonical

    seq = make_sequence(T=T, n_parts=8, pts_per_part=N // 8, seed=2, n_ref=3000, with_flow=use_flow)
    cano, pcs = split_canonical(seq["complete"], cano_idx)
    torch.manual_seed(seed)  # reference: --manual_seed 2 seeds the seg-head init
    model = BaseModel(num_parts=20, pose_len=T - 1).to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    refs = [t(r) for r in seq["ref_loc"]] if use_flow else None
    flows = [t(f) for f in seq["ref_flow"]] if use_flow else None
    eng = RelaxEngine(t(cano), t(pcs), model, cano_idx, refs, flows, n_iter=n_iter, seed=seed, use_grid=use_grid,
                      overlap_flow=overlap, profile=profile)
    return eng, seq, model


def cpu_baseline(seq, T, N, cano_idx, budget_s=20.0):
    """The oracle's iteration (C, OpenMP) on the host cores: a bounded sample of the same workload."""
    import oracle
    from oracle.step import RelaxOracle
    from reart_amd.synthetic import split_canonical

    cano, pcs = split_canonical(seq["complete"], cano_idx)
    rng = np.random.default_rng(0)
    H, P, B = 128, 20, T - 1
    W1 = rng.uniform(-0.5, 0.5, (H, 3)).astype(np.float32)
    b1 = rng.uniform(-0.5, 0.5, H).astype(np.float32)
    W2 = rng.uniform(-0.08, 0.08, (P, H)).astype(np.float32)
    p6d = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1))
    pt = np.zeros((B, P, 3), np.float32)
    orc = RelaxOracle(cano, pcs, W1, b1, W2, p6d, pt, cano_idx, seq.get("ref_loc"), seq.get("ref_flow"))
    noise = -np.log(rng.exponential(size=(N, P))).astype(np.float32)
    orc.step(noise)  # warm-up (page-in, thread pool)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.step(noise)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 20:
            break
    return {"value": n / el, "unit": "iterations/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"{n} iterations of the same step, oracle C/OpenMP, {el:.1f} s"}


def cpu_baseline_torch(seq, T, N, cano_idx, budget_s=10.0):
    """The reference-style PyTorch-CPU loop body (oracle/torch_step.py: the tensor expressions the reference issues,
    pinned to the reference's own trajectory golden) on the host cores: a bounded sample of the same workload."""
    from oracle.torch_step import TorchRelax
    from reart_amd.synthetic import split_canonical

    cano, pcs = split_canonical(seq["complete"], cano_idx)
    rng = np.random.default_rng(0)
    H, P, B = 128, 20, T - 1
    W1 = rng.uniform(-0.5, 0.5, (H, 3)).astype(np.float32)
    b1 = rng.uniform(-0.5, 0.5, H).astype(np.float32)
    W2 = rng.uniform(-0.08, 0.08, (P, H)).astype(np.float32)
    p6d = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1))
    pt = np.zeros((B, P, 3), np.float32)
    eng = TorchRelax(cano, pcs, W1, b1, W2, p6d, pt, cano_idx, seq.get("ref_loc"), seq.get("ref_flow"))
    eng.step()  # warm-up (thread pool, allocator)
    n, t0 = 0, time.perf_counter()
    while True:
        eng.step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 20:
            break
    return {"value": n / el, "unit": "iterations/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} iterations of the same step, reference-style PyTorch-CPU ops, {el:.1f} s"}


def extractor_flops(N, out_dim=64):
    """Useful multiply-add flops (x2) of ONE PointNet2Msg2 forward on a cloud of N points: every 1x1 conv of
    networks/feature_extractor.py:19-29 over the rows it is applied to (no tile padding counted)."""
    def stack(rows, cin, widths):
        f = 0
        for w in widths:
            f += 2 * rows * cin * w
            cin = w
        return f
    f = 0
    for K, widths in ((32, (32, 32, 64)), (64, (64, 64, 128)), (128, (64, 96, 128))):      # sa1: 512 centres, in = 3 + 3
        f += stack(512 * K, 6, widths)
    for K, widths in ((64, (128, 128, 256)), (128, (128, 196, 256))):                      # sa2: 128 centres, in = 320 + 3
        f += stack(128 * K, 323, widths)
    f += stack(128, 515, (256, 512, 1024))                                                 # sa3: group all
    f += stack(128, 1536, (256, 256))                                                      # fp3
    f += stack(512, 576, (256, 128))                                                       # fp2
    f += stack(N, 134, (128, 128))                                                         # fp1
    f += 2 * N * 128 * out_dim                                                             # conv1
    return f


def bench_extractor(args, dev):
    """The one-time correspondence extractor of BASELINE configs[2] at the loop's size: PointNet2Msg2 on the 2(T-1)
    clouds of N points that compute_corr_list_filter feeds it (utils/flow_utils.py:123-124).  One step = one forward of
    all clouds.  Weights: seeded (corr_model.pth.tar is not shipped), same as the parity goldens."""
    from reart_amd.networks.feature_extractor import PointNet2Msg2
    from reart_amd.synthetic import make_sequence
    from reart_amd.synthetic import extractor_state

    T, N = args.frames, args.points
    seq = make_sequence(T=T, n_parts=8, pts_per_part=N // 8, seed=2, with_flow=False)
    pts = torch.from_numpy(seq["complete"]).to(dev)
    pts = pts - pts.mean(dim=1, keepdim=True)
    pts = pts / pts.norm(dim=-1).max()
    xyz = torch.cat([pts[:-1], pts[1:]]).permute(0, 2, 1).contiguous()      # [2(T-1), 3, N]
    model = PointNet2Msg2(64)
    model.load_state_dict(extractor_state(model))
    model = model.to(dev).eval()
    B = xyz.shape[0]
    steps, warm = args.steps, args.warmup
    for _ in range(warm):
        f = model(xyz)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        f = model(xyz)
    ev1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = ev0.elapsed_time(ev1) / steps
    flops = extractor_flops(N) * B
    ach = flops / (ms * 1e-3) / 1e12
    # CPU leg: the reference itself cannot travel to the GPU box (in the build container its own PointNet2Msg2 takes 4.9 s
    # per 4096-point cloud on 8 threads).
    cpu = None
    if not getattr(args, "no_cpu_baseline", False):
        # the oracle's restatement of the whole forward (oracle/extractor.py: C sampling / grouping / interpolation + one
        # float32 matrix product per layer, pinned to the reference's own module by extractor.npz) on ONE cloud, bounded
        from oracle import extractor as ox
        import oracle as O_

        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        one = xyz[:1].cpu().numpy()
        ox.forward(sd, one, cuda_mode=True)                              # warm-up (thread pools)
        n_, t_ = 0, time.perf_counter()
        while True:
            ox.forward(sd, one, cuda_mode=True)
            n_ += 1
            e_ = time.perf_counter() - t_
            if e_ > getattr(args, "cpu_budget", 8.0) or n_ >= 20:
                break
        cpu = {"value": round(n_ / e_, 3), "unit": "clouds/s", "cores": max(O_.num_threads(), torch.get_num_threads()), "kind": "port",
               "sample": f"{n_} forwards of one {N}-point cloud, oracle PointNet2Msg2, {e_:.1f} s"}
    return {
        "metric": "correspondence-extractor clouds/sec", "value": round(B * steps / el, 2), "unit": "clouds/s", "n_gpus": 1,
        "steps": steps, "warmup": warm, "ms_per_step": round(1e3 * el / steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (fp32 MFMA, exact f32)", "data": "synthetic",
        "config": {"workload": f"PointNet2Msg2 forward on {B} clouds of {N} points (descriptors of a T={T} sequence)",
                   "clouds": B, "points": N, "sampling_rules": "CUDA (pointnet2_utils.CUDA = True)"},
        "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel": "mlp_gemm_kernel<NB> (v_mfma_f32_32x32x2_f32), whole forward's device time",
                     "kernel_ms": round(ms, 4), "useful_flops": flops},
        "cpu_baseline": cpu, "finite": bool(torch.isfinite(f).all()),
    }


def bench_kinematic(args, dev, rank, world, distributed, barrier):
    """BASELINE.json configs[4] as the reference runs it (README.md:125): kinematic projection from a relaxation result,
    `--model kinematic --use_flow_loss --use_assign_loss --assign_iter 0 --downsample 2 --assign_gap 1`, at the synthetic
    T x N of the headline.  One step = one iteration of run_robot.py:154-221 in that mode: forward kinematics of every
    frame, FPS subsets, (T-1) cost matrices of (N/2)^2, their OPTIMAL assignments (re-solved every iteration), assignment
    + flow loss, backward, Adam.  Returns the dict rank 0 prints."""
    from reart_amd import run_robot as rr
    from reart_amd import tail

    T, N = args.frames, args.points
    cano_idx = (T // 2 + rank) % T
    eng, seq, model = build_instance(dev, T, N, cano_idx, seed=2 + rank, use_flow=True)
    eng.capture(steps_per_graph=50)
    eng.step(args.base_iters)                       # the relaxation result the projection starts from (untimed)
    torch.cuda.synchronize()
    cano, pcs = eng.caller_clouds()
    with torch.no_grad():
        _, seg0, trans0 = model(cano)
    seg_s, trans_s, conn_s = tail.extract_structure(seg0, trans0, cano)
    result = {"pred_cano_part": seg_s.cpu().numpy(), "pred_pose_list": trans_s.cpu().numpy(),
              "joint_connection": conn_s.cpu().numpy().tolist(), "cano_idx": cano_idx}
    a = rr.build_parser().parse_args(["--model", "kinematic", "--use_flow_loss", "--use_assign_loss", "--assign_iter", "0",
                                      "--downsample", str(args.downsample), "--assign_gap", str(args.assign_gap),
                                      "--cano_idx", str(cano_idx), "--n_iter", "15000"])
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # build_graph prints like the reference does; stdout is ONE JSON line
        kin = rr.build_kinematic_from_base(result, cano, pcs, a).to(dev)
    t_ = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    refs, flows = [t_(r) for r in seq["ref_loc"]], [t_(f) for f in seq["ref_flow"]]
    loop = rr.make_projection_loop(a, kin, cano, pcs, refs, flows)      # the autograd-free KinematicEngine for this command
    it = 0
    for _ in range(args.warmup):
        loop.iteration(it); it += 1
    loop.lap_events = []
    barrier()
    solves0 = loop.lap_solves
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = loop.iteration(it); it += 1
    barrier()
    el = time.perf_counter() - t0
    if distributed:
        import torch.distributed as dist
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    if rank != 0:
        return None
    # dominant kernel: the assignment re-solve of the (T-1) problems, timed INSIDE the timed region with HIP events on the
    # launch stream around every solve (kernel + certificate + the host's copy of the result)
    from reart_amd.networks.pointnet2_utils import index_points
    from reart_amd.utils.lap import cdist, linear_sum_assignment_batch
    src_idx_b = loop.src_idx if loop.src_idx.dim() == 2 else loop.src_idx[None].expand(pcs.shape[0], -1)
    lap_ms = sum(e0.elapsed_time(e1) for e0, e1 in loop.lap_events) / max(len(loop.lap_events), 1)
    with torch.no_grad():
        pc_trans, _, _ = kin(cano)
        cold_src = index_points(pc_trans, src_idx_b).contiguous()
        cost = cdist(cold_src, loop.tgt_pts)
    n = cost.shape[1]
    ev0.record()
    # cold, as the loop's first refresh runs it: the epsilon-scaling auction, five schedules racing, chains from the points
    linear_sum_assignment_batch(cost, state={}, warm_assignment=True, points=(cold_src, loop.tgt_pts), race=True)
    ev1.record()
    torch.cuda.synchronize()
    lap_cold_ms = ev0.elapsed_time(ev1)
    # Roofline of the re-solve: a LATENCY bound.  A problem's re-solve is a sequential chain of steps (augmenting row
    # reduction + Dijkstra path search), each ending in a workgroup-wide arg-min over the problem's n column labels and one
    # barrier; neither HBM (both point sets live in LDS) nor the ALUs bound it.  floor = that arg-min + barrier alone, with
    # the solver's own primitives, measured live on this GPU (reart_lap_step_floor: T-1 workgroups like the solve);
    # achieved = the slowest problem's steps x floor = the time the chain cannot go below; frac = achieved / measured.
    from reart_amd import _lib as L_
    import ctypes
    floor_us = ctypes.c_double(0.0)
    fws = torch.empty(16 * cost.shape[0] + 256, dtype=torch.uint8, device=dev)
    L_.check(L_.lib().reart_lap_step_floor(cost.shape[0], n, 20000, L_.ptr(fws), fws.numel(), ctypes.byref(floor_us), L_.stream()),
             "reart_lap_step_floor")
    per_solve = np.asarray(getattr(loop, "lap_steps_log", [])[-max(len(loop.lap_events), 1):], dtype=np.float64)     # [solves, 3]: slowest problem (with backward rounds), mean, slowest problem's search steps
    steps_max = float(per_solve[:, 0].mean()) if per_solve.size else 0.0
    steps_mean = float(per_solve[:, 1].mean()) if per_solve.size else 0.0
    steps_search = float(per_solve[:, 2].mean()) if per_solve.size and per_solve.shape[1] > 2 else steps_max
    bound_ms = steps_max * floor_us.value * 1e-3
    solve_ms_all = [e0.elapsed_time(e1) for e0, e1 in loop.lap_events]
    # latency roofline: sequential workgroup-wide steps of the slowest problem x the measured floor of one
    # such step (arg-min + barrier, reart_lap_step_floor) over the measured solve; frac_search_only leaves the backward rounds out
    roof = {"bound": "latency", "achieved": round(bound_ms, 4), "peak": round(lap_ms, 4), "unit": "ms per re-solve (floor / measured)",
            "frac": round(bound_ms / lap_ms, 4) if lap_ms > 0 else None,
            "frac_search_only": round(steps_search * floor_us.value * 1e-3 / lap_ms, 4) if lap_ms > 0 else None, "traffic": None,
            "kernel": "lap_jvmw_kernel<32,2,16> + forest/arr/trees/set-up/passes (assignment re-solve)",
            "kernel_ms": round(lap_ms, 4), "solves_measured": len(loop.lap_events), **_pcts(solve_ms_all),
            "step_floor_us": round(floor_us.value, 4), "steps_slowest_problem": round(steps_max, 1),
            "search_steps_slowest_problem": round(steps_search, 1), "steps_mean_problem": round(steps_mean, 1),
            "cold_solve_ms": round(lap_cold_ms, 3)}
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import oracle
        cost_h = cost.cpu().numpy()
        pa, pb = index_points(pc_trans, src_idx_b).cpu(), loop.tgt_pts.cpu()
        t1 = time.perf_counter()
        c_cpu = torch.cdist(pa, pb).numpy()
        oracle.parallel_lap(c_cpu, nproc=len(c_cpu))
        el_cpu = time.perf_counter() - t1
        cpu = {"value": round(1.0 / el_cpu, 4), "unit": "iterations/s", "cores": min(len(cost_h), os.cpu_count() or 1),
               "kind": "reference",
               "sample": f"one refresh the reference's way: torch.cdist + scipy on {len(cost_h)} processes, {el_cpu:.1f} s"}
    return {
        "metric": "kinematic-projection iterations/sec", "value": round(world * args.steps / el, 3), "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (assignment potentials f64)",
        "data": "synthetic",
        "config": {"workload": f"kinematic projection (README.md:125), synthetic T={T} x N={N}, {T - 1} x {n}^2 per iteration",
                   "frames": T, "points": N,
                   "parts": int(trans_s.shape[1]), "assign_gap": args.assign_gap, "downsample": args.downsample,
                   "lap_solves_in_timed_region": loop.lap_solves - solves0 - 0, "lap_fallbacks": int(getattr(loop, "lap_fallbacks", -1)),
                   "parallelism": f"instances x{world}",
                   "loop": type(loop).__name__, "deterministic": bool(_ties_of(loop.lap_state) is not None), "ties": _ties_of(loop.lap_state)},
        "roofline": roof, "cpu_baseline": cpu,
        "final_losses": {k: float(v.detach()) for k, v in losses.items()},
    }


def run_secondary(args, dev, barrier):
    import copy

    keep = ("value", "unit", "steps", "ms_per_step", "roofline", "cpu_baseline", "cpu_baseline_torch")
    sec, start = {}, {}
    for name in ("kinematic", "extractor", "nao", "nao_recipe", "nao_projection"):
        a = copy.copy(args)
        a.cpu_budget = 3.0                                        # bounded CPU samples: the whole default run stays within minutes
        t0 = time.perf_counter()
        try:
            if name == "nao":
                a.steps = 15000                                   # the full 15 000 iterations of configs[2]
                full = bench_nao(a, dev)
            elif name == "nao_recipe":
                a.steps = 15000                                   # the README's 15 000 iterations with 2 000 refreshes
                full = bench_nao_recipe(a, dev, keep=start)
            elif name == "nao_projection":
                # README.md:125 from the recipe's result above: a bounded sample (the first 1 500 of its 15 000 iterations, three
                # windows); the whole run: `python bench.py --config nao_projection` (profiles/)
                full = bench_nao_projection(a, dev, start=start or None, n_iter=1500)
            elif name == "kinematic":
                a.steps, a.warmup = 100, 10                       # iterations 10-110 of the projection, like --config kinematic
                full = bench_kinematic(a, dev, 0, 1, False, barrier)
            else:
                a.steps, a.warmup = 20, 3
                full = bench_extractor(a, dev)
            sec[name] = {k: full[k] for k in keep if k in full}
            # the default line stays within the 5 KB the driver records: a secondary leg carries numbers; the labels of its
            # metric / kernel / units are in the full line of `python bench.py --config <name>` and in DESIGN.md section 6
            if isinstance(sec[name].get("roofline"), dict):
                drop = ("kernel", "unit", "traffic", "steps_slowest_problem", "search_steps_slowest_problem", "steps_mean_problem",
                        "sequential_steps_slowest_problem", "search_steps_mean_problem", "row_reduction_steps_mean_problem",
                        "backward_rounds_mean_problem")                    # (the step counts behind frac: in the full line of --config <name>)
                sec[name]["roofline"] = {k: v for k, v in sec[name]["roofline"].items() if k not in drop}
            for key in ("cpu_baseline", "cpu_baseline_torch"):
                if isinstance(sec[name].get(key), dict):
                    sec[name][key] = {k: v for k, v in sec[name][key].items() if k not in ("sample", "unit", "kind")}
                elif key == "cpu_baseline_torch":
                    sec[name].pop(key, None)
            extra = {"nao": ("matches_per_pair", "pairs_with_ground_truth_references", "correspondence_stage_s", "loop_s", "whole_run_s"),
                     "nao_recipe": ("correspondence_stage_s", "chamfer_phase_s", "assignment_phase_s", "loop_s", "whole_run_s",
                                    "assign_refreshes", "ms_per_refresh", "ms_per_solve", "first_solve_ms", "lap_fallbacks", "snapshots", "ties"),
                     "nao_projection": ("n_iter", "of", "snapshots", "iterations_per_s_by_window", "wall_s", "projected_whole_run_s",
                                        "lap_fallbacks", "deterministic", "ties", "other_mode"),
                     "kinematic": ("lap_fallbacks", "ties")}.get(name, ())
            sec[name].update({k: full["config"][k] for k in extra if k in full["config"]})
            if isinstance(sec[name].get("other_mode"), dict):            # the other mode of the projection: rate and percentiles only
                sec[name]["other_mode"] = {k: sec[name]["other_mode"][k] for k in ("deterministic", "value", "solve_ms_p50", "solve_ms_p95")}
            if isinstance(sec[name].get("ties"), dict):
                sec[name]["ties"] = [sec[name]["ties"]["flagged"], sec[name]["ties"]["changed"]]      # [problems flagged, problems changed]
        except Exception as exc:                                  # a secondary figure never costs the headline line
            sec[name] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        torch.cuda.synchronize()
        sec[name]["wall_s"] = round(time.perf_counter() - t0, 2)
    return sec


def nao_correspondences(dev):
    """The one-time stage of the nao runs (run_robot.py:64-84): descriptors of every frame -> mutual SMNN matches -> flow
    references; a frame pair with fewer than 3 matches (seeded extractor weights) takes 3 000 ground-truth correspondences.
    -> (data dict, cano, pc_list, cano_idx, complete [T,N,3], refs, flows, matches per pair, pairs on ground truth, seconds)."""
    from reart_amd.data import load_nao_demo
    from reart_amd.networks.feature_extractor import PointNet2Msg2
    from reart_amd.synthetic import extractor_state
    from reart_amd.utils.flow_utils import compute_corr_list_filter

    g = load_nao_demo()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cano, pcs, c = t(g["cano"]), t(g["pc_list"]), int(g["cano_idx"])
    complete = torch.cat((pcs[:c], cano[None], pcs[c:]), dim=0)                        # [T,N,3]
    T, N = complete.shape[:2]
    net = PointNet2Msg2(64)
    net.load_state_dict(extractor_state(net))
    net = net.to(dev).eval()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctr = complete.reshape(-1, 3).mean(0)
    norm = (complete - ctr) / (complete - ctr).norm(dim=-1).max()
    src_list, tgt_list = compute_corr_list_filter(norm, net, None, matching="smnn")
    refs = [complete[i][s_] for i, s_ in enumerate(src_list)]
    flows = [complete[i + 1][t_] - complete[i][s_] for i, (s_, t_) in enumerate(zip(src_list, tgt_list))]
    matches = [int(r.shape[0]) for r in refs]
    gt_pairs = 0
    if min(matches) < 3:      # too few mutual matches under random descriptors: ground-truth correspondences for that pair
        rng = np.random.default_rng(0)
        gt_pos, gt_flow = t(g["complete_gt_pc_list"]), t(g["gt_flow_list"])
        for i, m in enumerate(matches):
            if m < 3:
                sel = torch.from_numpy(rng.permutation(N)[:3000]).to(dev)
                refs[i], flows[i] = gt_pos[i][sel], gt_flow[i][sel]
                gt_pairs += 1
    torch.cuda.synchronize()
    return g, cano, pcs, c, complete, refs, flows, matches, gt_pairs, time.perf_counter() - t0


def bench_nao_recipe(args, dev, keep=None):
    """The relaxation recipe the reference's README documents for this sequence (README.md:116):
    `run_robot.py --seq_path=data/robot/nao --save_root=exp --cano_idx=2 --use_flow_loss --use_nproc --use_assign_loss
    --downsample 4 --n_iter=15000` -- with the defaults assign_iter = 5000, assign_gap = 5 (run_robot.py:386,404) that is
    5 000 iterations of Chamfer + flow loss, then 10 000 iterations of assignment + flow loss whose pairs are refreshed every 5
    iterations: 2 000 optimal assignments of 9 matrices of 1024 x 1024 (run_robot.py:164-187).  Correspondence stage as in
    `bench_nao`.  The timed whole run = correspondence stage + all 15 000 iterations + 2 000 refreshes."""
    from reart_amd import _lib as L_
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.run_robot import AssignmentPhase

    n_iter, assign_iter, gap, ds, lam = 15000, 5000, 5, 4, 0.3
    if args.steps != 15000:                                        # a shortened run for experiments: same proportions
        n_iter = max(args.steps, 30)
        assign_iter = n_iter // 3
    g, cano, pcs, c, complete, refs, flows, matches, gt_pairs, t_corr = nao_correspondences(dev)
    T, N = complete.shape[:2]
    B = T - 1
    torch.manual_seed(2)
    model = BaseModel(num_parts=20, pose_len=B).to(dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    eng = RelaxEngine(cano, pcs, model, c, refs, flows, n_iter=n_iter, seed=2)
    # the run a user starts prints a snapshot every --snapshot_gap = 100 iterations (run_robot.py:224-266: loss line + the
    # Flow / Seg / Recon eval lines on the ground truth the sequence carries): part of the timed run
    import functools
    import io

    from reart_amd.run_robot import SnapshotPrinter
    from reart_amd.utils.model_utils import tau_cosine
    snap_gap = 100
    tau_func = functools.partial(tau_cosine, max_iter=n_iter, end_temp=1.0, start_temp=5.0)
    sample = dict(gt_flow_list=g["gt_flow_list"], gt_cano_part=g["gt_cano_part"], complete_gt_pc_list=g["complete_gt_pc_list"])
    snap = SnapshotPrinter(argparse.Namespace(model="base", cano_idx=c), model, cano, pcs, sample, tau_func, out=io.StringIO())

    def snapshot(k, names):
        row = eng.last_losses().cpu().numpy()
        snap(k - 1, dict(zip(names, row[:3])))

    i = eng.capture(steps_per_graph=50)
    while i < assign_iter:
        chunk = min(snap_gap - i % snap_gap, assign_iter - i)
        eng.step(chunk)
        i += chunk
        if i % snap_gap == 0:
            snapshot(i, ("recon Loss", "flow Loss", "total Loss"))
    torch.cuda.synchronize()
    t_cd = time.perf_counter() - t1
    phase = AssignmentPhase(eng, cano, pcs, ds, gap, lam)
    phase.events, phase.collect_stats = [], True
    t2 = time.perf_counter()
    phase.run(i, n_iter, snap_gap, lambda k: snapshot(k, ("opt assignment loss", "flow Loss", "total Loss")))
    torch.cuda.synchronize()
    t_as = time.perf_counter() - t2
    rep = phase.report()
    n = phase.n
    if keep is not None:                                          # the relaxation result a projection can start from
        keep.update(model=model, cano=cano, pcs=pcs, cano_idx=c, refs=refs, flows=flows)
    # latency roofline of the refresh's solve, the re-solve's construction (see bench_kinematic): path-search steps of the
    # slowest problem x the measured floor of one workgroup-wide arg-min + barrier
    import ctypes
    floor_us = ctypes.c_double(0.0)
    fws = torch.empty(16 * B + 256, dtype=torch.uint8, device=dev)
    L_.check(L_.lib().reart_lap_step_floor(B, n, 20000, L_.ptr(fws), fws.numel(), ctypes.byref(floor_us), L_.stream()), "reart_lap_step_floor")
    st = np.asarray(phase.stats_log[1:], dtype=np.float64)          # [re-solves, 5]: see AssignmentPhase.stats_log
    steps_max = float(st[:, 0].mean()) if st.size else 0.0
    bound_ms = steps_max * floor_us.value * 1e-3
    solve_ms = rep.get("ms_per_solve", 0.0)
    cpu = None
    if not getattr(args, "no_cpu_baseline", False):
        import oracle
        from reart_amd.networks.pointnet2_utils import index_points

        eng.peek_forward()
        pa = index_points(eng.pc_trans, phase.src_idx.expand(B, n)).cpu()
        pb = phase.tgt_pts.cpu()
        tc = time.perf_counter()
        c_cpu = torch.cdist(pa, pb).numpy()
        oracle.parallel_lap(c_cpu, nproc=len(c_cpu))
        el_cpu = time.perf_counter() - tc
        cpu = {"value": round(gap / el_cpu, 3), "unit": "iterations/s", "cores": min(B, os.cpu_count() or 1), "kind": "reference",
               "sample": f"one refresh the reference's way: torch.cdist + scipy on {B} processes, {el_cpu:.2f} s per {gap} iterations"}
    whole = t_corr + t_cd + t_as
    return {
        "metric": "relaxation-recipe iterations/sec (README.md:116)", "value": round(n_iter / whole, 2), "unit": "iterations/s",
        "n_gpus": 1, "steps": n_iter, "warmup": 0, "ms_per_step": round(1e3 * whole / n_iter, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (assignment potentials f64)",
        "data": "reference demo sequence (nao), seeded extractor weights",
        "config": {"workload": f"nao relaxation as README.md:116 runs it, {rep['assign_refreshes']} refreshes of {B} x {n}^2",
                   "frames": T, "points": N, "n_iter": n_iter, "assign_iter": assign_iter, "assign_gap": gap, "downsample": ds,
                   "matches_per_pair": matches, "pairs_with_ground_truth_references": gt_pairs,
                   "correspondence_stage_s": round(t_corr, 4), "chamfer_phase_s": round(t_cd, 4), "assignment_phase_s": round(t_as, 4),
                   "loop_s": round(t_cd + t_as, 4), "whole_run_s": round(whole, 4), "assign_refreshes": rep["assign_refreshes"],
                   "ms_per_refresh": round(1e3 * t_as / max(rep["assign_refreshes"], 1), 4),
                   "ms_per_solve": round(solve_ms, 4), "first_solve_ms": round(rep.get("first_solve_ms", 0.0), 3),
                   "lap_fallbacks": rep["lap_fallbacks"], "snapshots": snap.count, "snapshot_gap": snap_gap,
                   "deterministic": bool(_ties_of(phase.lap_state) is not None), "ties": _ties_of(phase.lap_state)},
        "roofline": {"bound": "latency", "achieved": round(bound_ms, 4), "peak": round(solve_ms, 4),
                     "unit": "ms per re-solve (floor / measured)", "frac": round(bound_ms / solve_ms, 4) if solve_ms > 0 else None,
                     "frac_search_only": round(float(st[:, 3].mean()) * floor_us.value * 1e-3 / solve_ms, 4) if st.size and solve_ms > 0 else None,
                     "traffic": None, "kernel": "lap_jvmw_kernel<16,2,16> + forest/arr/trees/set-up/passes (assignment re-solve)",
                     "kernel_ms": round(solve_ms, 4), "solves_measured": max(len(phase.events) - 1, 0),
                     **_pcts([e0.elapsed_time(e1) for e0, e1 in phase.events[1:]]),
                     "step_floor_us": round(floor_us.value, 4), "sequential_steps_slowest_problem": round(steps_max, 1),
                     "search_steps_slowest_problem": round(float(st[:, 3].mean()) if st.size else 0.0, 1),
                     "backward_rounds_mean_problem": round(float(st[:, 4].mean()) if st.size else 0.0, 1),
                     "search_steps_mean_problem": round(float(st[:, 1].mean()) if st.size else 0.0, 1),
                     "row_reduction_steps_mean_problem": round(float(st[:, 2].mean()) if st.size else 0.0, 1)},
        "cpu_baseline": cpu, "final_losses": [float(v) for v in eng.last_losses().cpu()[:3]],
    }


def bench_nao_projection(args, dev, start=None, n_iter=None, windows=3):
    """The projection recipe the reference's README documents for this sequence (README.md:125): `run_robot.py --model=kinematic
    --use_flow_loss --use_assign_loss --assign_iter=0 --downsample=2 --assign_gap=1 --snapshot_gap=10 --base_result_path=...`
    on nao, from the result of the relaxation recipe (README.md:116; `start`: what bench_nao_recipe kept, or that recipe is
    run here first, untimed).  Every iteration re-solves 9 optimal assignments of 2048 x 2048 (run_robot.py:164-187) and every
    10th prints the reference's snapshot metrics (run_robot.py:224-266).  n_iter: the default line runs a BOUNDED sample -- the
    first `n_iter` iterations of the run, reported per window -- `--config nao_projection` without --steps = all 15 000."""
    import contextlib
    import io

    from reart_amd import run_robot as rr
    from reart_amd import tail

    if start is None:
        start = {}
        a0 = argparse.Namespace(**vars(args))
        a0.steps, a0.no_cpu_baseline = 15000, True
        bench_nao_recipe(a0, dev, keep=start)
    model, cano, pcs, c, refs, flows = (start[k] for k in ("model", "cano", "pcs", "cano_idx", "refs", "flows"))
    total = 15000
    n_iter = total if n_iter is None else int(n_iter)
    with torch.no_grad():
        _, seg0, trans0 = model(cano)
    seg_s, trans_s, conn_s = tail.extract_structure(seg0, trans0, cano)
    result = {"pred_cano_part": seg_s.cpu().numpy(), "pred_pose_list": trans_s.cpu().numpy(),
              "joint_connection": conn_s.cpu().numpy().tolist(), "cano_idx": c}
    a = rr.build_parser().parse_args(["--model", "kinematic", "--use_flow_loss", "--use_assign_loss", "--assign_iter", "0",
                                      "--downsample", "2", "--assign_gap", "1", "--snapshot_gap", "10", "--cano_idx", str(c),
                                      "--n_iter", str(total)])
    g = None
    try:
        from reart_amd.data import load_nao_demo
        g = load_nao_demo()
    except Exception:
        pass
    sample = None if g is None else dict(gt_flow_list=g["gt_flow_list"], gt_cano_part=g["gt_cano_part"],
                                         complete_gt_pc_list=g["complete_gt_pc_list"])
    from reart_amd.utils import lap as lap_
    edges = [round(n_iter * k / windows) for k in range(windows + 1)]

    def run_sample(deterministic):
        """The first n_iter iterations of the run from the recipe's result, tied optima settled canonically or not."""
        old = lap_.CANONICAL_TIES
        lap_.CANONICAL_TIES = bool(deterministic)
        try:
            with contextlib.redirect_stdout(sys.stderr):
                kin = rr.build_kinematic_from_base(result, cano, pcs, a).to(dev)
            loop = rr.make_projection_loop(a, kin, cano, pcs, refs, flows)
            snap = rr.SnapshotPrinter(a, kin, cano, pcs, sample, out=io.StringIO())        # the lines are produced, not shown
            loop.lap_events = []
            marks = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(n_iter):
                losses = loop.iteration(it)
                if it % a.snapshot_gap == 0 or it == n_iter - 1:
                    snap(it, losses)
                if it + 1 in edges[1:]:
                    torch.cuda.synchronize()
                    marks.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            tb = loop.lap_state.get("tie_breaker")
            return dict(loop=loop, kin=kin, snap=snap, marks=marks, el=el, losses=losses,
                        ms=[e0.elapsed_time(e1) for e0, e1 in loop.lap_events],
                        ties=None if tb is None else {"flagged": tb.flagged, "changed": tb.changed, "overflows": tb.overflows})
        finally:
            lap_.CANONICAL_TIES = old

    mode = bool(getattr(args, "deterministic", lap_.CANONICAL_TIES))
    run = run_sample(mode)
    loop, kin, snap, marks, el, losses, ms = (run[k] for k in ("loop", "kin", "snap", "marks", "el", "losses", "ms"))
    win = [round((edges[k + 1] - edges[k]) / (marks[k] - (marks[k - 1] if k else 0.0)), 2) for k in range(len(marks))]
    lap_ms = float(np.mean(ms[1:])) if len(ms) > 1 else 0.0
    n = loop.tgt_pts.shape[1]
    B = pcs.shape[0]
    # latency roofline of the re-solve, as bench_kinematic builds it: the slowest problem's sequential workgroup-wide steps x the
    # floor of one such step (arg-min + barrier) measured now, over the measured solve
    from reart_amd import _lib as L_
    import ctypes
    floor_us = ctypes.c_double(0.0)
    fws = torch.empty(16 * B + 256, dtype=torch.uint8, device=dev)
    L_.check(L_.lib().reart_lap_step_floor(B, n, 20000, L_.ptr(fws), fws.numel(), ctypes.byref(floor_us), L_.stream()), "reart_lap_step_floor")
    per_solve = np.asarray(getattr(loop, "lap_steps_log", [])[1:], dtype=np.float64)         # [re-solves, 3] (the cold solve left out)
    steps_max = float(per_solve[:, 0].mean()) if per_solve.size else 0.0
    steps_search = float(per_solve[:, 2].mean()) if per_solve.size else 0.0
    bound_ms = steps_max * floor_us.value * 1e-3
    # the other mode on the same bounded sample (the whole run is measured in one mode: profiles/)
    other = None
    if n_iter <= 3000 and not getattr(args, "one_mode", False):
        o = run_sample(not mode)
        o_ms = o["ms"]
        other = {"deterministic": not mode, "value": round(n_iter / o["el"], 3), "wall_s": round(o["el"], 3),
                 "kernel_ms": round(float(np.mean(o_ms[1:])) if len(o_ms) > 1 else 0.0, 4), **_pcts(o_ms[1:]),
                 "lap_fallbacks": int(o["loop"].lap_fallbacks), "ties": o["ties"]}
        del o
    cpu = None
    if not getattr(args, "no_cpu_baseline", False):
        # the reference's refresh on this run's own problems (run_robot.py:164-178: torch.cdist + parallel_lap over a pool,
        # utils/model_utils.py:85-89): ONE refresh of the 9 x 2048^2 on the host -- with --assign_gap 1 that is one iteration
        import oracle
        from reart_amd.networks.pointnet2_utils import index_points

        with torch.no_grad():
            pc_trans = kin(cano)[0]
        src_idx_b = loop.src_idx if loop.src_idx.dim() == 2 else loop.src_idx[None].expand(B, -1)
        pa, pb = index_points(pc_trans, src_idx_b).cpu(), loop.tgt_pts.cpu()
        t1 = time.perf_counter()
        c_cpu = torch.cdist(pa, pb).numpy()
        oracle.parallel_lap(c_cpu, nproc=len(c_cpu))
        el_cpu = time.perf_counter() - t1
        cpu = {"value": round(1.0 / el_cpu, 4), "unit": "iterations/s", "cores": min(B, os.cpu_count() or 1), "kind": "reference",
               "sample": f"one refresh the reference's way: torch.cdist + scipy on {B} processes, {el_cpu:.2f} s"}
    return {
        "metric": "kinematic-projection iterations/sec (README.md:125, nao)", "value": round(n_iter / el, 3), "unit": "iterations/s",
        "n_gpus": 1, "steps": n_iter, "warmup": 0, "ms_per_step": round(1e3 * el / n_iter, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (assignment potentials f64)",
        "data": "reference demo sequence (nao), seeded extractor weights",
        "config": {"workload": f"nao projection as README.md:125 runs it: first {n_iter} of {total} iterations, {B} x {n}^2 per iteration",
                   "n_iter": n_iter, "of": total, "snapshot_gap": a.snapshot_gap, "snapshots": snap.count,
                   "iterations_per_s_by_window": win, "wall_s": round(el, 3),
                   "projected_whole_run_s": round(el * total / n_iter, 1) if n_iter < total else round(el, 3),
                   "lap_fallbacks": int(loop.lap_fallbacks), "parts": int(trans_s.shape[1]),
                   "deterministic": mode, "ties": run["ties"], "other_mode": other},
        "roofline": {"bound": "latency", "achieved": round(bound_ms, 4), "peak": round(lap_ms, 4),
                     "unit": "ms per re-solve (floor / measured)", "frac": round(bound_ms / lap_ms, 4) if lap_ms > 0 else None,
                     "frac_search_only": round(steps_search * floor_us.value * 1e-3 / lap_ms, 4) if lap_ms > 0 else None,
                     "kernel": "lap_jvmw_kernel<32,2,16> + forest/arr/trees/set-up/passes (assignment re-solve)",
                     "kernel_ms": round(lap_ms, 4), "solves_measured": max(len(ms) - 1, 0), **_pcts(ms[1:]),
                     "step_floor_us": round(floor_us.value, 4), "steps_slowest_problem": round(steps_max, 1),
                     "search_steps_slowest_problem": round(steps_search, 1),
                     "first_solve_ms": round(ms[0], 3) if ms else None, "traffic": None},
        "cpu_baseline": cpu, "final_losses": {k: float(v.detach()) for k, v in losses.items()},
    }


def bench_nao(args, dev):
    """BASELINE.json configs[2]: the relaxation of the reference's demo sequence `nao` as its full loop runs it -- PointNet++
    descriptors of every frame -> mutual SMNN matches -> flow references (run_robot.py:64-84), then n_iter = 15 000
    iterations of run_robot.py:154-221 (per-part rigid transforms + Chamfer + flow loss, Adam) -- on one MI355X.  The nao
    clouds (10 frames x 4096 points, cano_idx 2) travel as DATA inside reart_amd/data/nao_demo.npz.  The reference does not
    ship `corr_model.pth.tar` / `category_normalize_scale.pkl`: the extractor runs on seeded weights with the clouds centred
    and scaled to the unit ball (timing is comparable, the quality of the matches is not); should a frame pair end with
    fewer than 3 matches, that pair's references come from the ground-truth flow (said in the line)."""
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine

    n_iter = args.steps
    g, cano, pcs, c, complete, refs, flows, matches, gt_pairs, t_corr = nao_correspondences(dev)
    T, N = complete.shape[:2]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    torch.manual_seed(2)
    model = BaseModel(num_parts=20, pose_len=T - 1).to(dev)
    eng = RelaxEngine(cano, pcs, model, c, refs, flows, n_iter=n_iter, seed=2, profile=True)
    spg = max(1, min(50, n_iter))
    done = eng.capture(steps_per_graph=spg)
    eng.step(min(150, max(n_iter - done, 0)))                      # warm-up inside the run, like the headline
    done = int(eng.iter.item())
    torch.cuda.synchronize()
    eng.search_profile(reset=True)
    t1 = time.perf_counter()
    eng.step(n_iter - done)
    torch.cuda.synchronize()
    el = time.perf_counter() - t1
    prof = eng.search_profile()
    steps = n_iter - done
    B = T - 1
    Mbar = float(np.mean([r.shape[0] for r in refs]))
    nn_pairs = 2 * B * N * N + B * N * Mbar
    k_ms = 1e3 * prof["seconds"] / max(prof["launches"], 1)
    executed = prof["pairs"] / max(prof["launches"], 1)
    ach = executed * 8 / (k_ms * 1e-3) / 1e12
    cpu = cpu_torch = None
    if not getattr(args, "no_cpu_baseline", False):
        import oracle
        from oracle.step import RelaxOracle

        rng = np.random.default_rng(0)
        H, P = 128, 20
        orc = RelaxOracle(g["cano"], g["pc_list"], rng.uniform(-0.5, 0.5, (H, 3)).astype(np.float32),
                          rng.uniform(-0.5, 0.5, H).astype(np.float32), rng.uniform(-0.08, 0.08, (P, H)).astype(np.float32),
                          np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1)), np.zeros((B, P, 3), np.float32), c,
                          [r.cpu().numpy() for r in refs], [f.cpu().numpy() for f in flows])
        noise = -np.log(rng.exponential(size=(N, P))).astype(np.float32)
        orc.step(noise)
        n, t2 = 0, time.perf_counter()
        while True:
            orc.step(noise); n += 1
            e2 = time.perf_counter() - t2
            if e2 > getattr(args, "cpu_budget", 8.0) or n >= 20:
                break
        cpu = {"value": round(n / e2, 3), "unit": "iterations/s", "cores": oracle.num_threads(), "kind": "port",
               "sample": f"{n} iterations of the same nao step, oracle C/OpenMP, {e2:.1f} s"}
        # the reference-style PyTorch-CPU loop body on the SAME nao step (networks/model.py:63-69,
        # utils/chamfer.py:174 as torch expressions: oracle/torch_step.py), bounded like the port's sample
        from oracle.torch_step import TorchRelax

        te = TorchRelax(g["cano"], g["pc_list"], rng.uniform(-0.5, 0.5, (H, 3)).astype(np.float32),
                        rng.uniform(-0.5, 0.5, H).astype(np.float32), rng.uniform(-0.08, 0.08, (P, H)).astype(np.float32),
                        np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (B, P, 1)), np.zeros((B, P, 3), np.float32), c,
                        [r.cpu().numpy() for r in refs], [f.cpu().numpy() for f in flows])
        te.step()
        n3, t3 = 0, time.perf_counter()
        while True:
            te.step(); n3 += 1
            e3 = time.perf_counter() - t3
            if e3 > getattr(args, "cpu_budget", 8.0) or n3 >= 20:
                break
        cpu_torch = {"value": round(n3 / e3, 3), "unit": "iterations/s", "cores": torch.get_num_threads(), "kind": "port",
                     "sample": f"{n3} iterations of the same nao step, reference-style PyTorch-CPU ops, {e3:.1f} s"}
    return {
        "metric": "relaxation-loop iterations/sec", "value": round(steps / el, 2), "unit": "iterations/s", "n_gpus": 1,
        "steps": steps, "warmup": done, "ms_per_step": round(1e3 * el / steps, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "reference demo sequence (nao), seeded extractor weights",
        "config": {"workload": f"nao relaxation, full loop (BASELINE configs[2]): descriptors + matches + {n_iter} iterations",
                   "frames": T, "points": N, "n_iter": n_iter,
                   "matches_per_pair": matches, "pairs_with_ground_truth_references": gt_pairs,
                   "correspondence_stage_s": round(t_corr, 4), "loop_s": round(el, 4),
                   "whole_run_s": round(t_corr + el, 4), "graph_replays": eng.graph_replays, "eager_steps": eng.eager_steps},
        "roofline": {"bound": "valu", "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": None, "kernel": "knn_group_kernel (as the headline's)",
                     "kernel_ms": round(k_ms, 5), "launches_measured": prof["launches"],
                     "executed_pairs_per_launch": round(executed, 1), "algorithmic_pairs_per_launch": int(nn_pairs),
                     "algorithmic_speedup": round(nn_pairs / max(executed, 1), 3)},
        "cpu_baseline": cpu, "cpu_baseline_torch": cpu_torch, "final_losses": [float(v) for v in eng.last_losses().cpu()[:3]],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="relax", choices=["relax", "kinematic", "extractor", "nao", "nao_recipe", "nao_projection"],
                    help="relax (default): BASELINE configs[1], the headline; kinematic: configs[4] (README.md:125); "
                         "extractor: the one-time PointNet++ correspondence extractor of configs[2]; nao: configs[2] itself -- the "
                         "reference's demo sequence, descriptors + matches + 15 000 iterations (--steps N for fewer); nao_recipe: the same "
                         "sequence as README.md:116 runs it (--use_flow_loss --use_assign_loss --downsample 4: 2 000 assignment refreshes); nao_projection: "
                         "README.md:125 on nao from that recipe's result, all 15 000 iterations with a snapshot every 10 (--steps N for the first N)")
    ap.add_argument("--base-iters", type=int, default=2000, help="kinematic: iterations of the relaxation the projection starts from")
    ap.add_argument("--assign-gap", type=int, default=1)
    ap.add_argument("--downsample", type=int, default=2)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps; default per --config: relax 1500, kinematic 100, extractor 20, nao / nao_recipe / nao_projection "
                         "the WHOLE run (15 000 iterations)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before; default per --config: relax 150, kinematic 10, extractor 3")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--no-flow", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--steps-per-graph", type=int, default=50,
                    help="iterations captured per graph (a graph launch costs ~8 us of idle GPU; the loop has no "
                         "host interaction, so several iterations replay as one graph)")
    ap.add_argument("--no-overlap", action="store_true", help="only affects the non-default search paths (brute force / grid): run their flow branch serially instead of on a second stream")
    ap.add_argument("--grid", action="store_true", help="exact grid search for the static targets (same results; slower at this size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the kinematic / extractor figures appended to the default line")
    ap.add_argument("--instances-per-gpu", type=int, default=1,
                    help="sweep mode: K independent instances share each GPU on separate streams (secondary figure; "
                         "the headline is K=1)")
    ap.add_argument("--no-tail", action="store_true", help="skip the end-of-run structure / energy timing (secondary figure)")
    ap.add_argument("--sweep-mode", choices=("batch", "streams"), default="batch",
                    help="how the sweep's instances share a GPU: batch = shared launches (reart_relax_step_batch), streams = one "
                         "stream per instance")
    ap.add_argument("--sweep-instances", type=int, default=6,
                    help="after the headline (one instance per GPU) also time this many concurrent instances per GPU "
                         "(--sweep-mode) and report the aggregate as `sweep` (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=20, help="eager steps timed per phase with HIP events")
    ap.add_argument("--deterministic", dest="deterministic", action="store_true", default=True,
                    help="(default, as in run_robot.py) assignment-bound configs: tied optima settled canonically")
    ap.add_argument("--no-deterministic", dest="deterministic", action="store_false")
    ap.add_argument("--one-mode", action="store_true", help="nao_projection: do not repeat the bounded sample in the other mode")
    args = ap.parse_args()
    # an explicit --steps N always means N (1500 used to be a sentinel for "the whole run" of the nao configs)
    args.full = args.steps is None
    from reart_amd.utils import lap as _lap

    _lap.CANONICAL_TIES = bool(args.deterministic)      # the product loops' default (run_robot.py / sweep: --deterministic)
    d_steps, d_warm = {"relax": (1500, 150), "kinematic": (100, 10), "extractor": (20, 3)}.get(args.config, (15000, 0))
    args.steps = d_steps if args.steps is None else args.steps
    args.warmup = d_warm if args.warmup is None else args.warmup

    world = _launch_module().check_world(args.gpus)      # --gpus N must be an N-rank job (or self-launched above)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ  # launched by torch.distributed.run (also with a single rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU (this node shows {torch.cuda.device_count()}); "
                         f"--gpus {args.gpus} needs {args.gpus} GPUs on one node")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        # RCCL prints a version banner to STDOUT when the communicator comes up; this program's stdout is ONE JSON
        # line, so stdout points at stderr while the communicator is created (init + a first collective)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == "kinematic":
        out = bench_kinematic(args, dev, rank, world, distributed, barrier)
        if rank == 0:
            print(_line(out))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.config in ("nao", "nao_recipe", "nao_projection"):
        if args.config == "nao_projection":
            out = bench_nao_projection(args, dev, n_iter=None if args.full else args.steps, windows=15 if args.full else 3)
        else:
            out = bench_nao(args, dev) if args.config == "nao" else bench_nao_recipe(args, dev)
        if rank == 0:
            print(_line(out))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.config == "extractor":
        out = bench_extractor(args, dev)
        if rank == 0:
            print(_line(out))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    T, N = args.frames, args.points
    use_flow = not args.no_flow
    # independent instances: rank r optimises canonical index (T//2 + r) % T (README: the
    # canonical frame is selected by the lowest final energy -> sweep over cano_idx)
    cano_idx = (T // 2 + rank) % T
    K = max(1, args.instances_per_gpu)
    engines, streams = [], []
    # a graph never holds more iterations than the timed region: a short driver run (--steps 20) still replays graphs
    spg = 1 if args.no_graph else max(1, min(args.steps_per_graph, args.steps))
    for k in range(K):
        st = torch.cuda.Stream(device=dev) if K > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            # profile=True: the search launch of every iteration leaves device-side stamps (2 stores per workgroup and
            # one extra workgroup in the consumer launch) -- this is how kernel_ms is measured INSIDE the timed region
            e_, seq, model = build_instance(dev, T, N, (cano_idx + k) % T, seed=2 + rank + 101 * k, use_flow=use_flow,
                                            use_grid=args.grid, overlap=not args.no_overlap, profile=(k == 0))
            used = 0 if args.no_graph else e_.capture(steps_per_graph=spg)
            e_.step(max(args.warmup - used, 0))
        engines.append(e_)
        streams.append(st)
    eng = engines[0]

    def run_steps(n):
        if K == 1:
            eng.step(n)
            return
        chunks = [spg] * (n // spg) + ([n % spg] if n % spg else [])
        for c in chunks:  # round-robin in whole graphs so that the instances interleave on the device
            for e_, st in zip(engines, streams):
                with torch.cuda.stream(st):
                    e_.step(c)

    barrier()
    prof = None
    try:
        eng.search_profile(reset=True)         # the accumulators cover exactly the timed region
    except Exception:                          # non-default search path (grid / brute force): no device-side profile
        pass
    for e_ in engines:
        e_.graph_replays = e_.eager_steps = 0
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    el = time.perf_counter() - t0
    try:
        prof = eng.search_profile()
    except Exception:
        prof = None
    replays, eager = eng.graph_replays, eng.eager_steps
    el_local = el                                   # this rank's own clock around the timed region (the line's `value` uses the max)
    if distributed:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    # secondary figure: the sweep the reference runs (one optimisation per canonical index) packs several
    # independent instances on a GPU; their launches interleave on separate streams and fill the issue slots
    # a single latency-bound instance leaves idle
    sweep = None
    if K == 1 and args.sweep_instances > 1 and args.sweep_mode == "batch":
        # the instances advance in SHARED launches (reart_relax_step_batch): every kernel of the iteration runs once with
        # one argument block per instance
        from reart_amd.relax import RelaxBatch

        Ks = args.sweep_instances
        sw_eng = [build_instance(dev, T, N, (cano_idx + 1 + k) % T, seed=1000 + rank + 101 * k, use_flow=use_flow,
                                 use_grid=args.grid, overlap=not args.no_overlap)[0] for k in range(Ks)]
        batch = RelaxBatch(sw_eng)
        # its own window, whatever --steps / --warmup say for the headline: the first ~150 iterations of an instance are the
        # start-up of the warm-started searches (every neighbour is new), and a 20-iteration window measures mostly that
        sw_warm, sw_steps = max(args.warmup, 150), max(args.steps, 300)
        sw_steps = (sw_steps + spg - 1) // spg * spg
        used = 0 if args.no_graph else batch.capture(steps_per_graph=spg)
        batch.step(max(sw_warm - used, 0))
        barrier()
        t1 = time.perf_counter()
        batch.step(sw_steps)
        barrier()
        el_s = time.perf_counter() - t1
        if distributed:
            tt = torch.tensor([el_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el_s = float(tt.item())
        sweep = {"instances_per_gpu": Ks, "value": round(world * Ks * sw_steps / el_s, 3), "unit": "iterations/s",
                 "per_gpu": round(Ks * sw_steps / el_s, 3), "n_gpus": world, "mode": "batch", "steps": sw_steps, "warmup": sw_warm,
                 "graph_replays": batch.graph_replays, "eager_steps": batch.eager_steps}
        del sw_eng, batch
    elif K == 1 and args.sweep_instances > 1:
        Ks = args.sweep_instances
        sw_eng, sw_st = [], []
        for k in range(Ks):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                e_, _, _ = build_instance(dev, T, N, (cano_idx + 1 + k) % T, seed=1000 + rank + 101 * k, use_flow=use_flow,
                                          use_grid=args.grid, overlap=not args.no_overlap)
                used = 0 if args.no_graph else e_.capture(steps_per_graph=spg)
                e_.step(max(args.warmup - used, 0))
            sw_eng.append(e_)
            sw_st.append(st)
        chunks = [spg] * (args.steps // spg) + ([args.steps % spg] if args.steps % spg else [])
        barrier()
        t1 = time.perf_counter()
        for c in chunks:
            for e_, st in zip(sw_eng, sw_st):
                with torch.cuda.stream(st):
                    e_.step(c)
        barrier()
        el_s = time.perf_counter() - t1
        if distributed:
            tt = torch.tensor([el_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el_s = float(tt.item())
        sweep = {"instances_per_gpu": Ks, "value": round(world * Ks * args.steps / el_s, 3), "unit": "iterations/s",
                 "per_gpu": round(Ks * args.steps / el_s, 3), "n_gpus": world, "mode": "streams"}
        del sw_eng
    losses = eng.last_losses()
    # gather of the final energies only (the reference's sweep picks argmin total energy)
    if distributed:
        gathered = [torch.zeros_like(losses) for _ in range(world)]
        dist.all_gather(gathered, losses)
        energies = torch.stack(gathered).cpu().numpy()
    else:
        energies = losses[None].cpu().numpy()
    # who ran what (one small object gather, outside the timed region): the driver's first multi-GPU run should be readable
    # ... and a straggler should be visible in the one line the driver keeps: every rank's own elapsed time and rate
    me = {"rank": rank, "device": str(dev), "cano_idx": int(cano_idx), "elapsed_s": round(el_local, 5),
          "it_per_s": round(K * args.steps / el_local, 2)}
    try:
        me["search_ms"] = round(1e3 * prof["seconds"] / prof["launches"], 5) if prof and prof["launches"] else None
    except Exception:
        me["search_ms"] = None
    if distributed:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    else:
        ranks_info = [me]

    # per-phase device time of the same step (eager, HIP events on the launch stream)
    phases = eng.step_timed(args.profile_steps) if args.profile_steps > 0 else {}
    if rank == 0:
        B = T - 1
        # Dominant kernel: the search launch.  With the flow loss it is ONE kernel (knn_pruned_pair_kernel)
        # holding the Chamfer K=1 search in both directions and the flow K=3 search.
        # Algorithmic bytes / flops per launch: every operand once, 8 flop per pair:
        #   Chamfer, per direction: read both clouds + write f32 dist + i64 idx; 2 B N^2 pairs
        #   flow: read queries and reference sets + write 3 x (f32 + i64);      B N M pairs
        M = 3000
        nn_bytes = 2 * (B * (N + N) * 12 + B * N * (4 + 8))
        nn_flops = 2 * B * N * N * 8
        kname = "knn_pruned_kernel<1> (Chamfer K=1 search, both directions)"
        if use_flow:
            nn_bytes += B * (N * 12 + M * 12 + N * 3 * (4 + 8))
            nn_flops += B * N * M * 8
            kname = "knn_pruned_pair_kernel (Chamfer K=1 both directions + flow K=3, one launch)"
        nn_pairs = nn_flops // 8
        roof = None
        if prof is not None and prof["launches"] > 0:
            # Measured INSIDE the timed region, in every iteration (graph replays included): per launch, the span from the
            # first workgroup's start to the last workgroup's end on the GPU's constant-rate clock, and the
            # query-target distance evaluations the launch actually executed (dense + sparse scans, seeds, rescans).
            k_ms = 1e3 * prof["seconds"] / prof["launches"]
            t_nn = k_ms * 1e-3
            executed = prof["pairs"] / prof["launches"]
            ach = executed * 8 / t_nn / 1e12
            traffic = traffic_current = None
            pmc_file = next((f for f in (os.path.join(ROOT, "profiles", "r06_pmc_search.json"), os.path.join(ROOT, "profiles", "r05_pmc_search.json"))
                             if os.path.exists(f)), None)
            if use_flow and T == 20 and N == 4096 and pmc_file:
                import hashlib

                pj = json.load(open(pmc_file))
                stamps = pj.get("measured_on_sources_sha256_16", {})
                same = bool(stamps) and all(hashlib.sha256(open(os.path.join(ROOT, p_), "rb").read()).hexdigest()[:16] == h_
                                            for p_, h_ in stamps.items())
                traffic = pj.get("hbm_bytes_per_launch")      # the committed rocprofv3 --pmc measurement (counters cannot be read inside a run)
                traffic_current = same                          # ... taken on the kernel sources this run uses?
            # kernel_ms and the executed pairs are device-side measurements of every launch in the timed region (profile stamps);
            # achieved = executed pair evaluations x 8 flop (no FMA: the distance contract) / kernel time against the fp32 vector
            # peak; the brute-force definition of SURVEY 8(d) is kept as algorithmic_* 
            roof = {"bound": "valu", "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_on_current_sources": traffic_current,
                    "kernel": "knn_group_kernel (Chamfer K=1 both directions + flow K=3, one launch)" if use_flow else "knn_group_kernel (Chamfer K=1, both directions)",
                    "kernel_ms": round(k_ms, 5), "launches_measured": prof["launches"],
                    "slow_box": bool(use_flow and T == 20 and N == 4096 and args.steps >= 300 and k_ms > 1.25 * 0.0329),
                    "executed_pairs_per_launch": round(executed, 1),
                    "algorithmic_pairs_per_launch": nn_pairs, "algorithmic_speedup": round(nn_pairs / executed, 3),
                    "algorithmic_frac": round(nn_flops / t_nn / 1e12 / FP32_PEAK_TFLOPS, 4),
                    "hbm_gbs": round(nn_bytes / t_nn / 1e9, 3), "algorithmic_bytes": nn_bytes}
            if use_flow and P_BENCH == 20:
                try:       # the five-launch chain with free arithmetic, measured now (an aid: never costs the line)
                    fl_us, fl_small = step_chain_floor(dev, N, B)
                    roof["step_floor_us"] = round(fl_us, 3)
                    roof["step_floor_small_kernels_us"] = round(fl_small, 3)
                    roof["step_floor_frac"] = round(fl_us / (1e6 * el / args.steps), 4)
                except Exception as exc:
                    roof["step_floor_us"] = None
        cpu = cpu_torch = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(seq, T, N, cano_idx)
            cpu_torch = cpu_baseline_torch(seq, T, N, cano_idx)
        # secondary figure: what ends an instance (reference run_robot.py:224-330) -- structure extraction and
        # the model-selection energy on the state reached above; outside the timed region, never part of `value`
        end_of_run = None
        if not args.no_tail:
            try:
                from reart_amd import tail

                def _timed(fn):
                    torch.cuda.synchronize()
                    t_ = time.perf_counter()
                    r_ = fn()
                    torch.cuda.synchronize()
                    return r_, 1e3 * (time.perf_counter() - t_)

                with torch.no_grad():
                    _, seg0, trans0 = eng.model(eng.cano)
                tail.extract_structure(seg0, trans0, eng.cano)                   # warm-up (lazy module / kernel load)
                (seg_s, trans_s, conn_s), ms_struct = _timed(lambda: tail.extract_structure(seg0, trans0, eng.cano))
                # warm-up of the energy's kernels (module load of the assignment solver: ~100 ms in a fresh process) on the
                # first 256 points; the timed call below is the full problem
                tail.energy_terms(eng.cano[:256].contiguous(), eng.pc_list[:, :256].contiguous(), seg_s[:256].contiguous(), trans_s, conn_s, cano_idx)
                en, ms_energy = _timed(lambda: tail.energy_terms(eng.cano, eng.pc_list, seg_s, trans_s, conn_s, cano_idx))
                cpu_tail = None
                if not args.no_cpu_baseline and world == 1:
                    # the same two stages on the host: the oracle's numpy restatement of the structure extraction and,
                    # for the energy, ONE of the T-1 assignments with scipy (what the reference calls); bounded sample
                    from oracle import structure as S_
                    from reart_amd.utils.lap import cdist as cdist_

                    seg_np, tr_np, cano_np = seg0.cpu().numpy(), trans0.detach().cpu().numpy(), eng.cano.cpu().numpy()
                    t_ = time.perf_counter()
                    dn_ = S_.denoise_seg_label(seg_np, cano_np, 20)
                    mg_ = S_.merging_wrapper(dn_, tr_np, cano_np, 3e-2, 2)
                    S_.extract_kinematic(mg_, tr_np, S_.mst_wrapper(mg_, tr_np, cano_np))
                    cpu_struct = 1e3 * (time.perf_counter() - t_)
                    from reart_amd.utils.model_utils import compute_pc_transform as cpt_
                    import oracle as O_

                    c1 = cdist_(cpt_(eng.cano, trans_s, seg_s)[:1], eng.pc_list[:1]).cpu().numpy()
                    t_ = time.perf_counter()
                    O_.linear_sum_assignment(c1)
                    cpu_lap = 1e3 * (time.perf_counter() - t_)
                    # (kind: port -- the oracle's structure extraction once; scipy on 1 of the T-1 matrices)
                    cpu_tail = {"structure_ms": round(cpu_struct, 1), "assignment_ms_per_matrix": round(cpu_lap, 1),
                                "matrices": int(eng.pc_list.shape[0])}
                end_of_run = {"structure_ms": round(ms_struct, 3), "energy_ms": round(ms_energy, 3), "cpu_baseline": cpu_tail,
                              "parts": int(trans_s.shape[1]), "total_err": round(en["total_err"], 6),
                              "ass_err": round(en["ass_err"], 6), "screw_err": round(en["screw_err"], 6),
                              "group_err": round(en["group_err"], 6), "lap_fallbacks": int(en.get("lap_fallbacks", -1))}
            except Exception as exc:      # a degenerate early state (e.g. every part merged) must not cost the bench line
                end_of_run = {"error": f"{type(exc).__name__}: {exc}"}
        out = {
            "metric": "relaxation-loop iterations/sec",
            "value": round(world * K * args.steps / el, 3),
            "unit": "iterations/s",
            "n_gpus": world,
            "per_gpu": round(K * args.steps / el, 3),
            "rccl_world": dist.get_world_size() if distributed else 1,
            "backend": dist.get_backend() if distributed else None,
            "ranks": ranks_info,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"synthetic T={T} x N={N}, P=20, Chamfer" + ("+flow" if use_flow else "") + ", full iteration, one instance/GPU",
                       "frames": T, "points": N, "parts": 20, "flow": use_flow,
                       "graph": not args.no_graph, "steps_per_graph": (0 if args.no_graph else spg),
                       "graph_replays": replays, "eager_steps": eager, "parallelism": f"instances x{world}" + (f" x{K} per GPU" if K > 1 else ""),
                       "instances_per_gpu": K},
            "roofline": roof,
            "cpu_baseline": cpu,
            "cpu_baseline_torch": cpu_torch,
            "sweep": sweep,
            "end_of_run": end_of_run,
            "phases_ms": {k: round(v, 5) for k, v in phases.items() if v},
            "final_losses": {"recon": float(energies[0][0]), "flow": float(energies[0][1]),
                             "per_rank_total": [float(e[2]) for e in energies]},
        }
        if world == 1 and not args.no_secondary:
            # BASELINE configs[4], the extractor of configs[2] and configs[2] itself (nao, 15 000 iterations) in the SAME line
            # (short in-process runs, a few seconds each),
            # each with its own roofline and CPU baseline; `python bench.py --config kinematic|extractor` gives the full lines
            out["secondary"] = run_secondary(args, dev, barrier)
        print(_line(out))
    if distributed:
        dist.barrier()   # rank 0 is still timing its secondary figures: nobody tears the communicator down early
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
