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

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector peak (counts FMA as 2)


def build_instance(dev, T, N, cano_idx, seed, use_flow=True, n_iter=15000, use_grid=False, overlap=True):
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
                      overlap_flow=overlap)
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
            "sample": f"{n} iterations of the same T={T} x N={N} Chamfer+flow step (oracle C/OpenMP, "
                      f"{el:.1f} s wall)"}


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
            "sample": f"{n} iterations of the same T={T} x N={N} Chamfer+flow step (reference-style PyTorch-CPU ops: "
                      f"conv1d / gumbel_softmax / bmm / cdist+argmin / topk / autograd / torch.optim.Adam, {el:.1f} s wall)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=150)
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
    ap.add_argument("--instances-per-gpu", type=int, default=1,
                    help="sweep mode: K independent instances share each GPU on separate streams (secondary figure; "
                         "the headline is K=1)")
    ap.add_argument("--no-tail", action="store_true", help="skip the end-of-run structure / energy timing (secondary figure)")
    ap.add_argument("--sweep-instances", type=int, default=3,
                    help="after the headline (one instance per GPU) also time this many concurrent instances per GPU on "
                         "separate streams and report the aggregate as `sweep` (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=20, help="eager steps timed per phase with HIP events")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ  # launched by torch.distributed.run (also with a single rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
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

    T, N = args.frames, args.points
    use_flow = not args.no_flow
    # independent instances: rank r optimises canonical index (T//2 + r) % T (README: the
    # canonical frame is selected by the lowest final energy -> sweep over cano_idx)
    cano_idx = (T // 2 + rank) % T
    K = max(1, args.instances_per_gpu)
    engines, streams = [], []
    for k in range(K):
        st = torch.cuda.Stream(device=dev) if K > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            e_, seq, model = build_instance(dev, T, N, (cano_idx + k) % T, seed=2 + rank + 101 * k, use_flow=use_flow,
                                            use_grid=args.grid, overlap=not args.no_overlap)
            used = 0 if args.no_graph else e_.capture(steps_per_graph=max(1, args.steps_per_graph))
            e_.step(max(args.warmup - used, 0))
        engines.append(e_)
        streams.append(st)
    eng = engines[0]

    def run_steps(n):
        if K == 1:
            eng.step(n)
            return
        spg = 1 if args.no_graph else max(1, args.steps_per_graph)
        chunks = [spg] * (n // spg) + ([n % spg] if n % spg else [])
        for c in chunks:  # round-robin in whole graphs so that the instances interleave on the device
            for e_, st in zip(engines, streams):
                with torch.cuda.stream(st):
                    e_.step(c)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    el = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    # secondary figure: the sweep the reference runs (one optimisation per canonical index) packs several
    # independent instances on a GPU; their launches interleave on separate streams and fill the issue slots
    # a single latency-bound instance leaves idle
    sweep = None
    if K == 1 and args.sweep_instances > 1:
        Ks = args.sweep_instances
        sw_eng, sw_st = [], []
        for k in range(Ks):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                e_, _, _ = build_instance(dev, T, N, (cano_idx + 1 + k) % T, seed=1000 + rank + 101 * k, use_flow=use_flow,
                                          use_grid=args.grid, overlap=not args.no_overlap)
                used = 0 if args.no_graph else e_.capture(steps_per_graph=max(1, args.steps_per_graph))
                e_.step(max(args.warmup - used, 0))
            sw_eng.append(e_)
            sw_st.append(st)
        spg = 1 if args.no_graph else max(1, args.steps_per_graph)
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
                 "note": "aggregate over all GPUs of Ks concurrent independent instances per GPU (separate streams); "
                         "`value` above is one instance per GPU"}
        del sw_eng
    losses = eng.last_losses()
    # gather of the final energies only (the reference's sweep picks argmin total energy)
    if distributed:
        gathered = [torch.zeros_like(losses) for _ in range(world)]
        dist.all_gather(gathered, losses)
        energies = torch.stack(gathered).cpu().numpy()
    else:
        energies = losses[None].cpu().numpy()

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
        roof = None
        if phases:
            try:   # back-to-back launches of the search alone between two HIP events on the launch stream
                k_ms, how = eng.search_ms(20), "mean of 20 back-to-back launches of the search kernel between two HIP events on the launch stream, at the optimisation state reached after the timed region"
            except RuntimeError:
                k_ms, how = phases["chamfer_nn"], "HIP events around the launch in eager serial steps after the timed region"
                if use_flow:
                    k_ms += phases["flow_knn3"]
            t_nn = k_ms * 1e-3
            ach = nn_bytes / t_nn / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_search.json")
            if os.path.exists(pmc) and use_flow:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            roof = {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic,
                    "kernel": kname, "kernel_ms": round(k_ms, 5), "algorithmic_bytes": nn_bytes,
                    "note": "the search is fp32-VALU bound by construction (>900 flop/B), see `valu`; kernel_ms: " + how,
                    "valu": {"achieved": round(nn_flops / t_nn / 1e12, 3), "peak": FP32_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(nn_flops / t_nn / 1e12 / FP32_PEAK_TFLOPS, 4),
                             "flops": nn_flops,
                             "note": "ALGORITHMIC flops (8 per query-target pair of the brute-force definition) over "
                                     "time; the exact box-pruned search evaluates only the pairs it cannot rule out, "
                                     "so this is work delivered, not ALU activity"}}
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
                    cpu_tail = {"structure_ms": round(cpu_struct, 1), "assignment_ms_per_matrix": round(cpu_lap, 1),
                                "matrices": int(eng.pc_list.shape[0]), "kind": "port",
                                "sample": "oracle (numpy, C k-NN) structure extraction once; scipy.optimize."
                                          "linear_sum_assignment on 1 of the T-1 matrices of 4096 x 4096 (serial)"}
                end_of_run = {"structure_ms": round(ms_struct, 3), "energy_ms": round(ms_energy, 3), "cpu_baseline": cpu_tail,
                              "parts": int(trans_s.shape[1]), "total_err": round(en["total_err"], 6),
                              "ass_err": round(en["ass_err"], 6), "screw_err": round(en["screw_err"], 6),
                              "group_err": round(en["group_err"], 6),
                              "note": "after warmup+steps iterations; energy_ms is dominated by the (T-1) optimal "
                                      "assignments of 4096 x 4096 (GPU auction + exact certificate)"}
            except Exception as exc:      # a degenerate early state (e.g. every part merged) must not cost the bench line
                end_of_run = {"error": f"{type(exc).__name__}: {exc}"}
        out = {
            "metric": "relaxation-loop iterations/sec",
            "value": round(world * K * args.steps / el, 3),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"synthetic articulated sequence T={T} x N={N}, P=20 parts, Chamfer"
                                   + ("+flow loss (k=3 blend, 3000 refs/pair)" if use_flow else " only")
                                   + ", full iteration fwd+loss+bwd+Adam, one instance per GPU",
                       "frames": T, "points": N, "parts": 20, "flow": use_flow,
                       "graph": not args.no_graph, "steps_per_graph": (0 if args.no_graph else max(1, args.steps_per_graph)), "grid_search_static_targets": args.grid, "parallelism": f"instances x{world}" + (f" x{K} per GPU" if K > 1 else ""),
                       "instances_per_gpu": K},
            "roofline": roof,
            "cpu_baseline": cpu,
            "cpu_baseline_torch": cpu_torch,
            "sweep": sweep,
            "end_of_run": end_of_run,
            "phases_ms": {k: round(v, 5) for k, v in phases.items()},
            "final_losses": {"recon": float(energies[0][0]), "flow": float(energies[0][1]),
                             "per_rank_total": [float(e[2]) for e in energies]},
        }
        print(json.dumps(out))
    if distributed:
        dist.barrier()   # rank 0 is still timing its secondary figures: nobody tears the communicator down early
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
