"""Synthetic articulated point-cloud sequence for the headline benchmark.

An articulated tree of 8 boxes (512 surface points each, N = 4096), side lengths U(0.04, 0.12),
scaled into [-0.35, 0.35]^3 (the range of the reference's nao demo data); T frames with
revolute joint angles theta_{t,e} = A_e sin(2 pi t / T + phi_e); every frame is sampled
independently on the surfaces (no point correspondence between frames, like the robot data
where each state_i.pkl is its own sample, reference dataset/dataset_robot.py:52-61).
Flow references: M points per pair drawn from frame i with flow = true motion + N(0, 0.002^2).
numpy only; deterministic in ``seed``.
"""
import numpy as np


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _box_points(rng, size, n):
    """n points uniform on the surface of an axis-aligned box centred at 0, as (face, u, v)."""
    areas = np.array([size[1] * size[2], size[0] * size[2], size[0] * size[1]] * 2)
    face = rng.choice(6, size=n, p=areas / areas.sum())
    uv = rng.uniform(-0.5, 0.5, (n, 2))
    pts = np.empty((n, 3))
    for f in range(6):
        m = face == f
        ax = f % 3
        o = [a for a in range(3) if a != ax]
        pts[m, ax] = (0.5 if f < 3 else -0.5) * size[ax]
        pts[m, o[0]] = uv[m, 0] * size[o[0]]
        pts[m, o[1]] = uv[m, 1] * size[o[1]]
    return pts


def make_sequence(T=20, n_parts=8, pts_per_part=512, seed=2, n_ref=3000, with_flow=True, amp_scale=1.0):
    """-> dict(complete [T,N,3] f32, part [N] i64, ref_idx/ref_loc/ref_flow lists for the T-1 pairs).
    ``amp_scale`` scales every joint's angular amplitude (hold-out sets of the solver's constants: tools/exp_tail.py)."""
    rng = np.random.default_rng(seed)
    sizes = rng.uniform(0.04, 0.12, (n_parts, 3))
    parent = [-1] + [int(rng.integers(0, i)) for i in range(1, n_parts)]
    # joint placement: child attached at a random face centre of the parent
    attach = rng.uniform(-0.5, 0.5, (n_parts, 3))
    axes = rng.normal(size=(n_parts, 3))
    amp = rng.uniform(0.2, 0.8, n_parts) * float(amp_scale)
    phi = rng.uniform(0, 2 * np.pi, n_parts)
    N = n_parts * pts_per_part

    def poses(t):
        Rw, tw = [None] * n_parts, [None] * n_parts
        for e in range(n_parts):
            if parent[e] < 0:
                Rw[e], tw[e] = np.eye(3), np.zeros(3)
                continue
            p = parent[e]
            ang = amp[e] * np.sin(2 * np.pi * t / T + phi[e])
            Rl = _rot(axes[e], ang)
            joint = attach[e] * sizes[p]                      # in the parent's frame
            child_off = np.array([0.0, 0.0, 0.5 * sizes[e][2]])  # child centre relative to the joint
            Rw[e] = Rw[p] @ Rl
            tw[e] = tw[p] + Rw[p] @ joint + Rw[e] @ child_off
        return Rw, tw

    def sample(t, local=None):
        Rw, tw = poses(t)
        pts = np.empty((N, 3))
        loc = [] if local is None else local
        for e in range(n_parts):
            if local is None:
                loc.append(_box_points(rng, sizes[e], pts_per_part))
            pts[e * pts_per_part:(e + 1) * pts_per_part] = loc[e] @ Rw[e].T + tw[e]
        return pts, loc

    frames, locals_ = [], []
    for t in range(T):
        pts, loc = sample(t)
        frames.append(pts)
        locals_.append(loc)
    allp = np.stack(frames)
    centre = 0.5 * (allp.reshape(-1, 3).max(0) + allp.reshape(-1, 3).min(0))
    scale = 0.35 / np.abs(allp - centre).max()
    complete = ((allp - centre) * scale).astype(np.float32)
    part = np.repeat(np.arange(n_parts), pts_per_part).astype(np.int64)
    out = dict(complete=complete, part=part, scale=scale)

    def world(t):   # part-local coordinates -> the scaled, centred frame of `complete`, as 4x4 per part
        Rw, tw = poses(t)
        W = np.tile(np.eye(4), (n_parts, 1, 1))
        for e in range(n_parts):
            W[e, :3, :3], W[e, :3, 3] = scale * Rw[e], scale * (tw[e] - centre)
        return W

    out["world"] = world
    if with_flow:
        ref_loc, ref_flow = [], []
        for t in range(T - 1):
            nxt, _ = sample(t + 1, local=locals_[t])         # the same material points, next frame
            nxt = ((nxt - centre) * scale)
            idx = rng.permutation(N)[: min(n_ref, N)]
            flow = nxt[idx] - complete[t][idx] + rng.normal(0, 0.002, (len(idx), 3))
            ref_loc.append(complete[t][idx].astype(np.float32))
            ref_flow.append(flow.astype(np.float32))
        out.update(ref_loc=ref_loc, ref_flow=ref_flow)
    return out


def split_canonical(complete, cano_idx):
    """-> (cano_pc [N,3], pc_list [T-1,N,3]) as reference dataset_robot.Sequence does (:88-89)."""
    return complete[cano_idx], np.concatenate([complete[:cano_idx], complete[cano_idx + 1:]], axis=0)


def export_sequence(path, T=6, n_parts=4, pts_per_part=256, seed=2, n_novel=2):
    """Write a generated sequence in the reference's on-disk layout (dataset/dataset_robot.py:9-45, utils/dataset_utils.py
    :15-26): ``state_i.pkl`` = {pc, part_id} for i = 0..T-1, ``pose_i.pkl`` = {part: 4x4 motion of the part from state 0
    to state i} for i >= 1, ``novel_pose_k.pkl`` = the same for poses between the frames.  -> the make_sequence dict."""
    import os
    import pickle

    seq = make_sequence(T=T, n_parts=n_parts, pts_per_part=pts_per_part, seed=seed, with_flow=False)
    os.makedirs(path, exist_ok=True)
    W0 = seq["world"](0)
    rel = lambda t: {int(e): seq["world"](t)[e] @ np.linalg.inv(W0[e]) for e in range(n_parts)}
    for t in range(T):
        with open(os.path.join(path, f"state_{t}.pkl"), "wb") as f:
            pickle.dump({"pc": seq["complete"][t].astype(np.float64), "part_id": seq["part"].copy()}, f)
        if t:
            with open(os.path.join(path, f"pose_{t}.pkl"), "wb") as f:
                pickle.dump(rel(t), f)
    for k in range(n_novel):
        with open(os.path.join(path, f"novel_pose_{k}.pkl"), "wb") as f:
            pickle.dump(rel(k + 1.5), f)
    return seq


def extractor_state(model, seed=11):
    """Deterministic weights for every parameter / buffer of a PointNet2Msg2-shaped module: the reference does not ship
    ``corr_model.pth.tar`` (feature_extractor.py:62-86), so the extractor benchmark and the parity fixtures run on these
    seeded weights (He-style scale keeps the activations O(1) through the 12 conv stacks)."""
    import torch

    rng = np.random.default_rng(seed)
    sd = {}
    for k, v in model.state_dict().items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(1, dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.normal(0, 0.1, shape).astype(np.float32))
        elif "bn" in k and k.endswith("weight"):
            sd[k] = torch.from_numpy(rng.uniform(0.8, 1.2, shape).astype(np.float32))
        elif k.endswith("bias"):
            sd[k] = torch.from_numpy(rng.normal(0, 0.05, shape).astype(np.float32))
        else:  # conv weight [out, in, 1(,1)]: He-style scale keeps activations O(1)
            fan_in = shape[1]
            sd[k] = torch.from_numpy(rng.normal(0, np.sqrt(2.0 / fan_in), shape).astype(np.float32))
    return sd
