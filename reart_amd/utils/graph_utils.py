"""Host-side mirror of the reference's ``utils/graph_utils.py`` (end-of-run structure extraction).

Same function names, argument meaning and return values as the reference; the numerical content runs in four
HIP kernels (``reart_amd/csrc/structure.hip``: ``reart_screw_fit``, ``reart_part_fps``, ``reart_part_pair_cost``,
``reart_group_temporal_err``) plus the existing search / transform operators.  What stays on the host is the
bookkeeping on a graph of at most ``num_parts`` (20) nodes: the greedy spanning tree over a P x P cost matrix and
the edge contraction -- a few microseconds of integer work after ONE device-to-host copy of the cost matrix
(the reference does a host sync per tree edge, utils/graph_utils.py:305-322, and builds networkx graphs).
No networkx here: the orderings networkx would produce (Kahn generations, adjacency insertion order) are
written out, and the tests compare them with networkx through the oracle.
"""
import numpy as np
import torch

from .. import _lib
from .model_utils import compute_pc_transform, knn_query


# ------------------------------------------------------------------------------------------------ kernels
def screw_fit(trans_list, pairs=None, plain_mean=False, want=("cost",)):
    """``reart_screw_fit``: trans_list [T,P,4,4] + pairs [E,2] (src, tgt), or relative motions [T,E,4,4] with
    ``pairs=None`` -> dict with the requested entries of: ``screw`` [T,E,8], ``rel`` [T,E,4,4], ``mean`` [E,6],
    ``recon`` [T,E,4,4], ``cost`` [E,4] (revolute, prismatic, min, identity), ``mean_cost`` scalar tensor."""
    _lib.require_gpu(trans_list)
    tr = trans_list.detach().contiguous().float()
    dev = tr.device
    T = tr.shape[0]
    if pairs is None:
        E, P, pp = tr.shape[1], tr.shape[1], None
    else:
        pp = torch.as_tensor(pairs, device=dev).to(torch.int32).contiguous()
        E, P = pp.shape[0], tr.shape[1]
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    out = {"cost": new(E, 4), "mean": new(E, 6)}
    if "screw" in want:
        out["screw"] = new(T, E, 8)
    if "rel" in want:
        out["rel"] = new(T, E, 4, 4)
    if "recon" in want:
        out["recon"] = new(T, E, 4, 4)
    if "mean_cost" in want:
        out["mean_cost"] = new(1)
    g = lambda k: _lib.ptr(out[k]) if k in out else None
    rc = _lib.lib().reart_screw_fit(_lib.ptr(tr), T, P, _lib.ptr(pp) if pp is not None else None, E, int(plain_mean),
                                    g("screw"), g("rel"), g("mean"), g("recon"), g("cost"), g("mean_cost"), None, 0,
                                    _lib.stream())
    _lib.check(rc, "reart_screw_fit")
    return out


def _pair_cost(cano_fps, frame_fps=None):
    _lib.require_gpu(cano_fps)
    cf = cano_fps.detach().contiguous().float()
    Ps, F = cf.shape[:2]
    dev = cf.device
    ff = frame_fps.detach().contiguous().float() if frame_fps is not None else None
    T = ff.shape[0] if ff is not None else 0
    dist = torch.empty((Ps, Ps), dtype=torch.float32, device=dev)
    pair = torch.empty((Ps, Ps, 2), dtype=torch.int64, device=dev)
    joint = torch.empty((Ps, Ps), dtype=torch.float32, device=dev) if ff is not None else None
    rc = _lib.lib().reart_part_pair_cost(_lib.ptr(cf), _lib.ptr(ff) if ff is not None else None, T, Ps, F,
                                         _lib.ptr(dist), _lib.ptr(pair), _lib.ptr(joint) if joint is not None else None,
                                         _lib.stream())
    _lib.check(rc, "reart_part_pair_cost")
    return dist, pair, joint


# ------------------------------------------------------------------------------------ reference interface
def fps_sample_cano(cano_pc, cano_part, uni_label, num_fps=20, cuda_mode=None):
    """utils/graph_utils.py:37-52: farthest point sampling inside every part of ``uni_label`` (one launch).
    -> (points [P,num_fps,3], indices into cano_pc [P,num_fps]).  Raises ValueError for a part below num_fps.
    Sampling rules: ``cuda_mode`` None follows ``networks.pointnet2_utils.CUDA`` (both start at the part's first
    member here; the CPU fallback's random start is pinned to 0 as in the fixtures)."""
    from ..networks import pointnet2_utils as _pn2

    _lib.require_gpu(cano_pc, cano_part)
    cuda_mode = _pn2._rules(cuda_mode)
    cano = cano_pc.detach().contiguous().float()
    seg = cano_part.contiguous().long()
    lab = torch.as_tensor(uni_label, device=cano.device).long().contiguous()
    Ps = lab.shape[0]
    idx = torch.empty((Ps, num_fps), dtype=torch.int64, device=cano.device)
    cnt = torch.empty((Ps,), dtype=torch.int32, device=cano.device)
    rc = _lib.lib().reart_part_fps(_lib.ptr(cano), _lib.ptr(seg), cano.shape[0], _lib.ptr(lab), Ps, num_fps,
                                   int(cuda_mode), _lib.ptr(idx), _lib.ptr(cnt), _lib.stream())
    _lib.check(rc, "reart_part_fps")
    small = torch.nonzero(cnt < num_fps)
    if small.numel():
        k = int(small[0])
        raise ValueError("part id {} too small, only {} points".format(int(lab[k]), int(cnt[k])))
    return cano[idx], idx


def fps_index_list(pc_trans_list, cano_part_idx_list):
    """utils/graph_utils.py:55-67: [T,N,3], [P,num_fps] -> [T,P,num_fps,3]."""
    return pc_trans_list[:, cano_part_idx_list]


def compute_spatial_cost(cano_part_fps_list, chamfer_dist=None, return_index=False):
    """utils/graph_utils.py:70-84: closest-pair squared distance between the FPS sets of every ordered part pair
    (``chamfer_dist`` is accepted for signature parity; the pair search is its own kernel)."""
    dist, pair, _ = _pair_cost(cano_part_fps_list)
    return (dist, pair) if return_index else dist


def compute_joint_cost(part_fps_list, joint_connection, edge_pair_indices):
    """utils/graph_utils.py:87-100: squared distance between the chosen FPS point of part ``joint_connection[e, 0]``
    and that of part ``joint_connection[e, 1]``, per frame.  part_fps_list [T,P,F,3] (or [P,F,3]) -> [T,E] (or [E])."""
    a = part_fps_list[..., joint_connection[:, 0], edge_pair_indices[:, 0], :]
    b = part_fps_list[..., joint_connection[:, 1], edge_pair_indices[:, 1], :]
    return (a - b).square().sum(dim=-1)


def _part_counts(cano_part):
    lab, cnt = torch.unique(cano_part, sorted=True, return_counts=True)
    return lab, cnt


def filter_seg_label(cano_part, min_num=10):
    """utils/graph_utils.py:103-112."""
    lab, cnt = _part_counts(cano_part)
    return lab[cnt >= min_num]


def denoise_seg_label(cano_part, cano_pc, knn, min_num=10):
    """utils/graph_utils.py:115-123: parts below ``min_num`` points take the label of the nearest kept point
    (in place, like the reference)."""
    lab, cnt = _part_counts(cano_part)
    mask = torch.isin(cano_part, lab[cnt < min_num])
    if bool(mask.any()):
        cano_part[mask] = knn_query(cano_pc[mask].contiguous(), cano_pc[~mask].contiguous(), cano_part[~mask], knn)
    return cano_part


def compute_relative_trans(trans_list, return_trans=False):
    """utils/graph_utils.py:170-186: screw parameters of inv(T_i) T_j for all ordered part pairs and frames."""
    T, P = trans_list.shape[:2]
    ar = torch.arange(P, device=trans_list.device)
    pairs = torch.stack([ar.repeat_interleave(P), ar.repeat(P)], dim=1)
    f = screw_fit(trans_list, pairs, want=("screw", "rel"))
    s = f["screw"].reshape(T, P, P, 8)
    res = (s[..., 0:3], s[..., 3:6], s[..., 6], s[..., 7])
    return res + (f["rel"].reshape(T, P, P, 4, 4),) if return_trans else res


def compute_geo_cost(rel_trans, axis=None, moment=None, theta=None, distance=None):
    """utils/graph_utils.py:131-167: min(revolute, prismatic) reconstruction cost of every part pair [P,P]
    (the screw parameters are recomputed from ``rel_trans`` inside the kernel; the extra arguments are accepted
    for signature parity)."""
    T, P = rel_trans.shape[:2]
    f = screw_fit(rel_trans.reshape(T, P * P, 4, 4))
    return f["cost"][:, 2].reshape(P, P)


def frobenius_cost(predict, gt):
    """utils/graph_utils.py:189-196 (host-side torch: used by callers outside the tail only)."""
    R, t = gt[:, :3, :3], gt[:, :3, 3:]
    igt = torch.zeros_like(gt)
    igt[:, :3, :3] = R.transpose(1, 2)
    igt[:, :3, 3:] = -R.transpose(1, 2) @ t
    igt[:, 3, 3] = 1.0
    err = predict @ igt - torch.eye(4, dtype=predict.dtype, device=predict.device)
    return (err * err).sum(dim=(-2, -1))


def compute_root_cost(trans_list):
    """utils/graph_utils.py:199-203: distance of every part's motion from the identity [P]."""
    eye = torch.eye(4, dtype=trans_list.dtype, device=trans_list.device)
    return ((trans_list - eye) ** 2).sum(dim=(2, 3)).mean(dim=0)


def compute_screw_trans(trans_list, return_cost=False):
    """utils/graph_utils.py:235-283: trans_list [T,E,4,4] relative motions -> reconstruction with the cheaper joint
    type per edge [T,E,4,4] (and mean_e(cost) / T)."""
    f = screw_fit(trans_list, want=("recon", "mean_cost"))
    return (f["recon"], f["mean_cost"][0]) if return_cost else f["recon"]


def compute_screw_cost(pred_trans_list, pred_connection):
    """utils/graph_utils.py:286-292."""
    f = screw_fit(pred_trans_list, pred_connection, want=("mean_cost",))
    return f["mean_cost"][0]


def mst(cost, uni_label=None, max_cost=None, keep_index=False, verbose=False):
    """utils/graph_utils.py:295-324: greedy spanning tree -- repeatedly the first minimum (row-major) of
    cost + 1e10 * [same component], in fp32.  One device-to-host copy of ``cost``; the loop is host integer work."""
    c = cost.detach().to(torch.float32).cpu().numpy()
    n = c.shape[0]
    lab = None if (uni_label is None or keep_index) else [int(x) for x in torch.as_tensor(uni_label).tolist()]
    comp = list(range(n))
    big = np.float32(1e10)
    out = []
    for _ in range(n - 1):
        same = np.equal.outer(np.asarray(comp), np.asarray(comp))
        cur = c + np.where(same, big, np.float32(0))
        k = int(cur.argmin())
        i, j = divmod(k, n)
        if max_cost is not None and cur[i, j] > max_cost:
            break
        if verbose:
            print(i if lab is None else lab[i], j if lab is None else lab[j], float(cur[i, j]))
        ci, cj = comp[i], comp[j]
        comp = [ci if x == cj else x for x in comp]
        out.append([i, j] if lab is None else [lab[i], lab[j]])
    dev = cost.device
    return torch.tensor(out, dtype=torch.long, device=dev).reshape(-1, 2)


def _kahn_generations(nodes, succ):
    """Topological order by generations (zero in-degree nodes in insertion order, successors in insertion order)."""
    indeg = {v: 0 for v in nodes}
    for u in nodes:
        for v in succ[u]:
            indeg[v] += 1
    gen = [v for v in nodes if indeg[v] == 0]
    order = []
    while gen:
        nxt = []
        for u in gen:
            order.append(u)
            for v in succ[u]:
                indeg[v] -= 1
                if indeg[v] == 0:
                    nxt.append(v)
        gen = nxt
    if len(order) != len(nodes):
        raise ValueError("There are cycles in the link graph")
    return order


def contract_edges(edges, cost, merge_thr, verbose=False):
    """The graph part of merge_graph (utils/graph_utils.py:344-385) on plain lists: ``edges`` [[a, b], ...] of a
    directed tree, ``cost`` per edge.  Nodes are visited in topological order (generations of zero in-degree nodes,
    ascending labels first); each out-edge the node has WHEN IT IS REACHED is contracted if its cost is below
    ``merge_thr`` (edges inherited during this visit wait for the next merging round).
    -> (relabel {old label: surviving label}, remaining edges in node / adjacency order)."""
    nodes = sorted({int(x) for e in edges for x in e})
    succ = {v: [] for v in nodes}      # ordered out-adjacency: [target, cost]
    pred = {v: [] for v in nodes}
    for (a, b), c in zip(edges, cost):
        succ[a].append([b, c])
        pred[b].append(a)
        if verbose:
            print("add edge {}-{}: cost {}".format(a, b, c))
    topo = _kahn_generations(nodes, {v: [t for t, _ in succ[v]] for v in nodes})
    alive = set(nodes)
    relabel = {v: v for v in nodes}
    for node in topo:
        if node not in alive:
            continue
        for tgt, c in list(succ[node]):
            if tgt in alive and any(e[0] == tgt for e in succ[node]) and c < merge_thr:
                # contract tgt into node: the edge disappears, tgt's other edges move to node / to its predecessors,
                # each re-inserted at the END of the adjacency it lands in
                succ[node] = [e for e in succ[node] if e[0] != tgt]
                for p2 in pred[tgt]:
                    if p2 == node:
                        continue
                    moved = [e for e in succ[p2] if e[0] == tgt]
                    succ[p2] = [e for e in succ[p2] if e[0] != tgt]
                    if all(e[0] != node for e in succ[p2]):
                        succ[p2].append([node, moved[0][1]])
                        pred[node].append(p2)
                for t2, c2 in succ[tgt]:
                    pred[t2] = [x for x in pred[t2] if x != tgt]
                    if t2 != node and all(e[0] != t2 for e in succ[node]):
                        succ[node].append([t2, c2])
                        pred[t2].append(node)
                pred[node] = [x for x in pred[node] if x != tgt]
                alive.discard(tgt)
                for k, v in relabel.items():
                    if v == tgt:
                        relabel[k] = node
                if verbose:
                    print("merge edge {}-{}: cost {}".format(tgt, node, c))
    remaining = [[a, t] for a in nodes if a in alive for t, _ in succ[a]]
    und = {v: set() for v in alive}
    for a, b in remaining:
        und[a].add(b)
        und[b].add(a)
    seen, stack = set(), [next(iter(alive))]
    while stack:
        v = stack.pop()
        if v not in seen:
            seen.add(v)
            stack.extend(und[v] - seen)
    if seen != alive:
        raise ValueError("New graph are not all connected.")
    _kahn_generations([v for v in nodes if v in alive], {v: [t for t, _ in succ[v]] for v in alive})
    return relabel, remaining


def merge_graph(seg_part, joint_connection, trans_list, merge_thr, verbose=True):
    """utils/graph_utils.py:327-385: contract every tree edge whose relative motion stays within ``merge_thr`` of
    the identity.  -> (merged labels, remaining edges [E',2])."""
    dev = seg_part.device
    cost = screw_fit(trans_list, joint_connection)["cost"][:, 3].cpu().tolist()    # mean_t |inv(T_a) T_b - I|^2
    relabel, remaining = contract_edges(joint_connection.cpu().tolist(), cost, merge_thr, verbose)
    lut = torch.arange(max(int(seg_part.max()), max(relabel)) + 1, device=dev)
    for k, v in relabel.items():
        lut[k] = v
    return lut[seg_part], torch.tensor(remaining, dtype=joint_connection.dtype, device=dev).reshape(-1, 2)


def _pair_costs_for(seg_part, trans_list, cano_pc, uni_label, num_fps, pred_pc_list=None):
    if pred_pc_list is None:
        pred_pc_list = compute_pc_transform(cano_pc, trans_list, seg_part)
    fps_pts, fps_idx = fps_sample_cano(cano_pc, seg_part, uni_label, num_fps=num_fps)
    return _pair_cost(fps_pts, fps_index_list(pred_pc_list, fps_idx))


def merging_wrapper(seg_part, trans_list, cano_pc, chamfer_dist, merge_thr, n_it=2):
    """utils/graph_utils.py:388-416: ``n_it`` rounds of (candidate tree over closest-pair + joint-drift costs,
    contraction of the near-rigid edges)."""
    pred_pc_list = compute_pc_transform(cano_pc, trans_list, seg_part)
    for _ in range(n_it):
        uni_label = torch.unique(seg_part, sorted=True)
        cano_dist, _, joint_cost = _pair_costs_for(seg_part, trans_list, cano_pc, uni_label, 20, pred_pc_list)
        merge_cost = cano_dist + joint_cost
        merge_cost = merge_cost + 1e4 * torch.eye(merge_cost.shape[0], device=merge_cost.device, dtype=merge_cost.dtype)
        candidates = mst(merge_cost, uni_label=uni_label)
        seg_part, _ = merge_graph(seg_part, candidates, trans_list, merge_thr, verbose=False)
        if not len(torch.unique(seg_part)) > 1:
            break
    return seg_part


def mst_wrapper(seg_part, trans, cano_pc, chamfer_dist, verbose=False, num_fps=20, cano_dist_thr=1e-2,
                joint_cost_weight=100):
    """utils/graph_utils.py:419-447: the kinematic tree = spanning tree over
    [closest pair too far] * 1e4 + screw reconstruction cost + joint_cost_weight * joint drift."""
    uni_label = torch.unique(seg_part, sorted=True)
    Ps = uni_label.shape[0]
    pairs = torch.stack([uni_label.repeat_interleave(Ps), uni_label.repeat(Ps)], dim=1)
    geo_cost = screw_fit(trans, pairs)["cost"][:, 2].reshape(Ps, Ps)
    cano_dist, _, joint_cost = _pair_costs_for(seg_part, trans, cano_pc, uni_label, num_fps)
    dist_cost = 0 * (cano_dist < cano_dist_thr) + 1e4 * (cano_dist >= cano_dist_thr)
    cost = dist_cost + geo_cost + joint_cost_weight * joint_cost
    cost = cost + 1e4 * torch.eye(Ps, device=cost.device, dtype=cost.dtype)
    return mst(cost, uni_label=uni_label, verbose=verbose)
