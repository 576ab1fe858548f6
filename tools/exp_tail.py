#!/usr/bin/env python3
"""Where the assignment re-solve's time goes REFRESH BY REFRESH (mean 1.9 ms, max 42 ms per search launch).
Runs the README recipe's assignment phase on the nao demo (MODE=recipe: 9 x 1024^2 every 5 iterations, run_robot.py:164-187) or
the kinematic projection that follows it (MODE=projection: README.md:125, 9 x 2048^2 every iteration), times every solve with
HIP events, keeps the solver's per-problem statistics and the state every solve started from, and prints
  * percentiles of the solve time, * how the time follows the statistics, * the slowest solves with their statistics;
DUMP=path.npz stores the inputs of the KEEP slowest solves and of SAMPLE solves evenly spaced over the run (source points,
previous columns and potentials; targets once) so that a solver variant can be replayed on exactly those problems
(tools/replay_tail.py: the loops' trajectories are chaotic, two runs of a loop never compare the same problems).
Usage: gpurun -- 'MODE=recipe ITERS=15000 python tools/exp_tail.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

MODE = os.environ.get("MODE", "recipe")
KEEP = int(os.environ.get("KEEP", 24))
DUMP = os.environ.get("DUMP", "")
SAMPLE = int(os.environ.get("SAMPLE", 0))      # DUMP also keeps this many solves evenly spaced over the run (the typical ones)
dev = torch.device("cuda:0")


def pct(a, q):
    return float(np.percentile(np.asarray(a, dtype=np.float64), q)) if len(a) else float("nan")


def table(ms, raw, form_note):
    """ms [S], raw [S][B,4] solver statistics per solve (lap_jvmw_kernel's o[0..3], unmasked)."""
    ms = np.asarray(ms)
    raw = np.asarray(raw)
    freed = raw[:, :, 0] & 0xffff
    back = (raw[:, :, 0] >> 21) & 0x3ff
    left = raw[:, :, 1] & 0xffff
    steps = raw[:, :, 2]
    arr = raw[:, :, 3] >> 8
    print(f"solves {len(ms)} ({form_note}): mean {ms.mean():.3f} ms | p50 {pct(ms, 50):.3f} | p90 {pct(ms, 90):.3f} | p95 {pct(ms, 95):.3f} | "
          f"p99 {pct(ms, 99):.3f} | max {ms.max():.3f} | share of the time in the slowest 5 % of the solves: "
          f"{np.sort(ms)[-max(len(ms) // 20, 1):].sum() / ms.sum():.2f}")
    seq = (steps + back).max(axis=1)
    print(f"per solve, slowest problem: search steps p50 {pct(steps.max(1), 50):.0f} p95 {pct(steps.max(1), 95):.0f} max {steps.max():.0f} | "
          f"rows left p50 {pct(left.max(1), 50):.0f} p95 {pct(left.max(1), 95):.0f} max {left.max()} | rows released p50 {pct(freed.max(1), 50):.0f} "
          f"max {freed.max()} | row-reduction steps (problem mean) {arr.mean():.0f}")
    if os.environ.get("BUCKET_STATS"):      # diagnostic build (-DMW_BUCKET_STATS): relaxations / 16 | rounds | buckets in place of conflicts / reduction steps
        relax = 16 * ((raw[:, :, 1] >> 16) & 0xffff)
        rounds, buckets = arr & 0xfff, (arr >> 12) & 0xfff
        k = steps.argmax(1)
        ar = np.arange(len(ms))
        print(f"bucketed part, slowest problem of a solve: relaxations p50 {pct(relax[ar, k], 50):.0f} p95 {pct(relax[ar, k], 95):.0f} max {relax.max()} | "
              f"rounds p50 {pct(rounds[ar, k], 50):.0f} p95 {pct(rounds[ar, k], 95):.0f} | buckets p50 {pct(buckets[ar, k], 50):.0f} p95 {pct(buckets[ar, k], 95):.0f} | "
              f"totals over all problems: relaxations {relax.sum()} rounds {rounds.sum()} buckets {buckets.sum()} counted steps {steps.sum()}")
        ok = (rounds[ar, k] < 4000) & (buckets[ar, k] < 4000) & (ms < np.percentile(ms, 99))
        seqs = np.maximum(steps[ar, k] - rounds[ar, k] - buckets[ar, k], 0)
        A_ = np.stack([np.ones(ok.sum()), seqs[ok], rounds[ar, k][ok], buckets[ar, k][ok], relax[ar, k][ok]], axis=1).astype(np.float64)
        c_, *_ = np.linalg.lstsq(A_, ms[ok], rcond=None)
        print(f"least squares over the slowest problem's counts: ms = {c_[0]:.3f} + {1e3 * c_[1]:.3f} us x one-column steps + {1e3 * c_[2]:.3f} us x rounds + "
              f"{1e3 * c_[3]:.3f} us x buckets + {1e3 * c_[4]:.3f} us x relaxations; residual rms {(ms[ok] - A_ @ c_).std():.3f} ms; "
              f"means: steps {seqs[ok].mean():.0f} rounds {rounds[ar, k][ok].mean():.0f} buckets {buckets[ar, k][ok].mean():.0f} relaxations {relax[ar, k][ok].mean():.0f}")
    # a linear account: ms ~ a + b * (search steps of the slowest problem)
    A = np.stack([np.ones(len(ms)), steps.max(1)], axis=1)
    coef, *_ = np.linalg.lstsq(A, ms, rcond=None)
    res = ms - A @ coef
    print(f"least squares ms = {coef[0]:.3f} + {1e3 * coef[1]:.3f} us x (search steps of the slowest problem); residual rms {res.std():.3f} ms, "
          f"largest residuals {np.sort(res)[-5:].round(2).tolist()}")
    print("slowest solves: index | ms | per problem: released / left / search steps / backward rounds / reduction steps")
    order = np.argsort(-ms)[:KEEP]
    for s in order[:12]:
        print(f"  {s:6d} | {ms[s]:7.3f} | " + " ".join(f"{freed[s, b]}/{left[s, b]}/{steps[s, b]}/{back[s, b]}/{arr[s, b]}" for b in range(raw.shape[1])))
    if SAMPLE:
        order = np.concatenate([order, np.linspace(0, len(ms) - 1, SAMPLE).astype(np.int64)])
    return order, seq


def main():
    from reart_amd.data import load_nao_demo
    from reart_amd.networks.model import BaseModel
    from reart_amd.relax import RelaxEngine
    from reart_amd.run_robot import AssignmentPhase

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n_iter, assign_iter = int(os.environ.get("ITERS", 15000)), int(os.environ.get("ASSIGN_ITER", 5000))
    ds_recipe, ds_proj = int(os.environ.get("DS", 4)), int(os.environ.get("DS_PROJ", 2))
    if os.environ.get("SEQ", "").startswith("synthetic:"):
        # a HOLD-OUT sequence (the solver's constants were tuned on nao alone):
        # SEQ=synthetic:seed,parts,points per part,frames,amplitude scale -- reart_amd.synthetic.make_sequence, canonical frame T // 2
        from reart_amd.synthetic import make_sequence, split_canonical
        sd, parts, ppp, T, amp = os.environ["SEQ"].split(":", 1)[1].split(",")
        seq = make_sequence(T=int(T), n_parts=int(parts), pts_per_part=int(ppp), seed=int(sd), n_ref=3000, with_flow=True, amp_scale=float(amp))
        c = int(T) // 2
        cano_h, pcs_h = split_canonical(seq["complete"], c)
        cano, pcs = t(cano_h), t(pcs_h)
        refs, flows = [t(r) for r in seq["ref_loc"]], [t(f) for f in seq["ref_flow"]]
    else:
        g = load_nao_demo()
        cano, pcs, c = t(g["cano"]), t(g["pc_list"]), int(g["cano_idx"])
        rng = np.random.default_rng(0)
        gt_pos = t(g["complete_gt_pc_list"])
        sel = [torch.from_numpy(rng.permutation(gt_pos.shape[1])[:3000]).to(dev) for _ in range(pcs.shape[0])]
        refs = [gt_pos[k][s] for k, s in enumerate(sel)]
        flows = [t(g["gt_flow_list"][k])[s] for k, s in enumerate(sel)]
    torch.manual_seed(2)
    model = BaseModel(num_parts=20, pose_len=pcs.shape[0]).to(dev)
    eng = RelaxEngine(cano, pcs, model, c, refs, flows, n_iter=n_iter, seed=2)
    i = eng.capture(steps_per_graph=50)
    eng.step(assign_iter - i)
    phase = AssignmentPhase(eng, cano, pcs, ds_recipe, 5, 0.3)
    phase.events, phase.collect_stats = [], True
    before, srcs = [], []
    orig = phase._refresh_on_device

    def hooked():
        st = phase.lap_state
        before.append((st["cols"].clone(), st["prices"].clone()))
        fb = orig()
        srcs.append(phase._src_pts.clone())
        return fb

    if MODE == "recipe":
        phase._refresh_on_device = hooked
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    phase.run(assign_iter, n_iter)
    torch.cuda.synchronize()
    print(f"recipe assignment phase: {time.perf_counter() - t0:.2f} s, {phase.refreshes} refreshes, fallbacks {phase.fallbacks}")
    if MODE == "recipe":
        ms = [a.elapsed_time(b) for a, b in phase.events][-len(before):]
        raw = phase.stats_raw[-len(before):]
        order, _ = table(ms, raw, f"recipe {pcs.shape[0]} x {phase.n}^2, reart_lap_resolve_points_mc")
        if DUMP:
            np.savez_compressed(DUMP, tgt=phase.tgt_pts.cpu().numpy(), idx=order, ms=np.asarray(ms)[order],
                                src=np.stack([srcs[s].cpu().numpy() for s in order]),
                                cols=np.stack([before[s][0].cpu().numpy() for s in order]),
                                prices=np.stack([before[s][1].cpu().numpy() for s in order]))
        return
    # ---- the projection that follows (README.md:125), from this run's result
    import contextlib
    from reart_amd import run_robot as rr
    from reart_amd import tail
    with torch.no_grad():
        _, seg0, trans0 = model(cano)
    seg_s, trans_s, conn_s = tail.extract_structure(seg0, trans0, cano)
    result = {"pred_cano_part": seg_s.cpu().numpy(), "pred_pose_list": trans_s.cpu().numpy(),
              "joint_connection": conn_s.cpu().numpy().tolist(), "cano_idx": c}
    a = rr.build_parser().parse_args(["--model", "kinematic", "--use_flow_loss", "--use_assign_loss", "--assign_iter", "0",
                                      "--downsample", str(ds_proj), "--assign_gap", "1", "--cano_idx", str(c)])
    with contextlib.redirect_stdout(sys.stderr):
        kin = rr.build_kinematic_from_base(result, cano, pcs, a).to(dev)
    loop = rr.make_projection_loop(a, kin, cano, pcs, refs, flows)
    p_iter = int(os.environ.get("P_ITERS", 3000))
    loop.lap_events = []
    raw, bef, srcp = [], [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for it in range(p_iter):
        st = loop.lap_state
        if st.get("cols") is not None and st.get("prices") is not None:
            bef.append((st["cols"].clone(), st["prices"].clone()))
        loop.iteration(it)
        if len(bef) > len(srcp):
            srcp.append(loop.pc_trans[:, loop.src_idx].clone())
            r_ = np.array(loop.lap_stats)
            # lap.py takes the backward rounds out of word 0 (state["backward_rounds"]): back where table() reads them
            r_[:, 0] = (r_[:, 0] & 0x1fffff) | (np.asarray(loop.lap_state.get("backward_rounds", np.zeros(len(r_), np.int64))).astype(np.int32) << 21)
            raw.append(r_)
        if (it + 1) % 500 == 0:
            torch.cuda.synchronize()
            marks.append((it + 1, time.perf_counter() - t0))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"projection: {p_iter} iterations in {el:.2f} s = {p_iter / el:.1f} it/s; fallbacks {loop.lap_fallbacks}; per 500: "
          + " ".join(f"{500 / (b[1] - (marks[k - 1][1] if k else 0.0)):.0f}" for k, b in enumerate(marks)))
    if getattr(loop, "lap_winners", None) is not None:           # raced re-solves: which racer (order of the free rows) finished first, per problem and solve
        print("racer that finished first (problems x solves):", " ".join(str(int(v)) for v in loop.lap_winners[:14]))
    ms = [x.elapsed_time(y) for x, y in loop.lap_events][-len(bef):]
    raw = np.asarray(raw)
    order, _ = table(ms, raw, f"projection {pcs.shape[0]} x {loop.tgt_pts.shape[1]}^2")
    if DUMP:
        np.savez_compressed(DUMP, tgt=loop.tgt_pts.cpu().numpy(), idx=order, ms=np.asarray(ms)[order],
                            src=np.stack([srcp[s].cpu().numpy() for s in order]),
                            cols=np.stack([bef[s][0].cpu().numpy() for s in order]),
                            prices=np.stack([bef[s][1].cpu().numpy() for s in order]))


if __name__ == "__main__":
    main()
