"""CPU restatement (numpy, fp32) of the reference's end-of-run structure extraction -- TEST INFRASTRUCTURE ONLY
(imported by tests/, never by the product).  Pinned by tests/golden/structure.npz, which the reference's own
functions produced here (tests/golden/make_golden_structure.py).

Citations are into the reference tree: ``gu`` = utils/graph_utils.py, ``ku`` = utils/kinematic_utils.py,
``mu`` = utils/model_utils.py, ``dq`` = screw_se3/dq_utils.py, ``geo`` = screw_se3/geo_utils.py.
Graph bookkeeping (topological order, shortest paths, edge contraction) uses networkx exactly where the
reference does: like scipy's linear_sum_assignment it is the reference's own third-party call.
"""
import math

import numpy as np

import oracle as _o

F32 = np.float32


def inverse_transformation(T):
    """geo:9-53: [R|t] -> [R^T | -R^T t]."""
    T = np.asarray(T, F32)
    out = np.zeros_like(T)
    Rt = np.swapaxes(T[..., :3, :3], -1, -2)
    out[..., :3, :3] = Rt
    out[..., :3, 3:4] = np.matmul(-Rt, T[..., :3, 3:4])
    out[..., 3, 3] = 1.0
    return out


def matrix_to_quaternion(M):
    """geo:536-587 (best-conditioned of the four candidates, real part first)."""
    M = np.asarray(M, F32).reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [M[:, i] for i in range(9)]
    one = F32(1.0)
    arg = np.stack([one + m00 + m11 + m22, one + m00 - m11 - m22, one - m00 + m11 - m22, one - m00 - m11 + m22], -1)
    q_abs = np.where(arg > 0, np.sqrt(np.maximum(arg, 0)), F32(0)).astype(F32)
    cand = np.stack([
        np.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        np.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], -1),
        np.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], -1),
        np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], -1)], -2).astype(F32)
    cand = cand / (F32(2.0) * np.maximum(q_abs[:, :, None], F32(0.1)))
    return cand[np.arange(M.shape[0]), q_abs.argmax(-1)].astype(F32)


def _q_mul(q1, q2):
    """dq:63-83."""
    w1, x1, y1, z1 = [q1[:, i] for i in range(4)]
    w2, x2, y2, z2 = [q2[:, i] for i in range(4)]
    w = w2 * w1 - x2 * x1 - y2 * y1 - z2 * z1
    x = w2 * x1 + x2 * w1 - y2 * z1 + z2 * y1
    y = w2 * y1 + x2 * z1 + y2 * w1 - z2 * x1
    z = w2 * z1 - x2 * y1 + y2 * x1 + z2 * w1
    return np.stack([w, x, y, z], 1).astype(F32)


def transform_to_screw(T, eps=1e-6):
    """dq:137-182 after dq:129-134: [n,4,4] -> (l [n,3], m [n,3], theta [n], d [n])."""
    T = np.asarray(T, F32).reshape(-1, 4, 4)
    n = T.shape[0]
    q_r = matrix_to_quaternion(T[:, :3, :3])
    tq = np.concatenate([np.zeros((n, 1), F32), T[:, :3, 3]], 1)
    q_d = (F32(0.5) * _q_mul(tq, q_r)).astype(F32)
    # rotation angle (dq:99-111)
    qn = (q_r / np.sqrt((q_r * q_r).sum(-1, dtype=F32))[:, None]).astype(F32)
    nim = np.sqrt((qn[:, 1:] ** 2).sum(-1, dtype=F32)).astype(F32)
    theta = (F32(2.0) * np.arctan2(nim, qn[:, 0])).astype(F32)
    no_rot = (np.abs(theta) < eps) | (np.abs(theta - F32(math.pi)) < eps)
    conj = q_r * np.array([1, -1, -1, -1], F32)
    dq_t = _q_mul(F32(2.0) * q_d, conj)[:, 1:]
    l = np.zeros((n, 3), F32)
    d = np.zeros(n, F32)
    wr = ~no_rot
    l[wr] = q_r[wr, 1:] / np.sin(theta[wr] / F32(2.0))[:, None]
    d[no_rot] = np.sqrt((dq_t[no_rot] ** 2).sum(-1, dtype=F32))
    l[no_rot] = dq_t[no_rot] / (d[no_rot, None] + F32(1e-10))
    cos = l.sum(-1, dtype=F32)
    theta = np.where(cos >= 0, theta, -theta)
    l = np.where(cos[:, None] >= 0, l, -l)
    d[no_rot] = np.where(cos[no_rot] >= 0, d[no_rot], -d[no_rot])
    d[wr] = (dq_t[wr] * l[wr]).sum(-1, dtype=F32)
    unit = no_rot & np.isclose(d, 0)
    l[unit, 0] = 1
    theta = theta.copy()
    theta[no_rot] = eps
    tl = np.cross(dq_t, l).astype(F32)
    m = (F32(0.5) * (tl + np.cross(l, tl / np.tan(theta / F32(2.0))[:, None]))).astype(F32)
    return l.astype(F32), m, theta.astype(F32), d.astype(F32)


def mean_screw_param(s_axis, moment, theta, distance, eps_tol=1e-5):
    """gu:207-232: [T,E,3] x2, [T,E] x2 -> mean axis / moment [E,3] over the frames that are not a unit transform."""
    T, E = s_axis.shape[:2]
    if E <= 1:
        return s_axis.mean(0, dtype=F32), moment.mean(0, dtype=F32)
    no_rot = (np.abs(theta) <= eps_tol) | (np.abs(theta - F32(math.pi)) <= eps_tol)
    unit = no_rot & (distance <= eps_tol)
    ma, mm = np.empty((E, 3), F32), np.empty((E, 3), F32)
    for e in range(E):
        keep = np.ones(T, bool) if unit[:, e].all() else ~unit[:, e]
        ma[e], mm[e] = s_axis[keep, e].mean(0, dtype=F32), moment[keep, e].mean(0, dtype=F32)
    return ma, mm


def frobenius_cost(pred, gt):
    """gu:189-196: sum of squares of pred * gt^-1 - I."""
    err = np.matmul(np.asarray(pred, F32), inverse_transformation(gt)) - np.eye(4, dtype=F32)
    return (err * err).sum((-2, -1), dtype=F32)


def screw_fit(rel):
    """The common body of gu:compute_geo_cost (:131-167) and gu:compute_screw_trans (:235-283) on relative
    transforms rel [T,E,4,4]: screw parameters per frame, their masked means, the revolute (d = 1e-6) and the
    prismatic (theta = 1e-6, rotation := I) reconstructions and their costs."""
    rel = np.asarray(rel, F32)
    T, E = rel.shape[:2]
    l, m, th, d = transform_to_screw(rel.reshape(-1, 4, 4))
    l, m, th, d = l.reshape(T, E, 3), m.reshape(T, E, 3), th.reshape(T, E), d.reshape(T, E)
    ma, mm = mean_screw_param(l, m, th, d)
    mae, mme = np.broadcast_to(ma, (T, E, 3)).reshape(-1, 3), np.broadcast_to(mm, (T, E, 3)).reshape(-1, 3)
    tiny = np.full(T * E, 1e-6, F32)
    rec_r = _o.screw_to_transform(mae, mme, th.reshape(-1), tiny).reshape(T, E, 4, 4)
    cost_r = frobenius_cost(rec_r, rel).sum(0, dtype=F32)
    rel_p = rel.copy()
    rel_p[..., :3, :3] = np.eye(3, dtype=F32)
    rec_p = _o.screw_to_transform(mae, mme, tiny, d.reshape(-1)).reshape(T, E, 4, 4)
    cost_1 = frobenius_cost(rec_p, rel_p).sum(0, dtype=F32)
    cost_2 = F32(((rec_p[..., :3, :3] - rel[..., :3, :3]).astype(np.float64) ** 2).mean())
    cost_p = cost_1 + cost_2
    return dict(axis=l, moment=m, theta=th, distance=d, mean_axis=ma, mean_moment=mm, recon_r=rec_r, recon_p=rec_p,
                cost_r=cost_r, cost_p=cost_p, cost=np.minimum(cost_r, cost_p))


def relative_trans(trans, src, tgt):
    """inv(trans[:, src]) @ trans[:, tgt] -> [T,E,4,4] (gu:170-175, gu:286-290, ku:87-89)."""
    trans = np.asarray(trans, F32)
    return np.matmul(inverse_transformation(trans[:, src]), trans[:, tgt]).astype(F32)


def geo_cost(trans, labels):
    """gu:compute_relative_trans + compute_geo_cost restricted to ``labels`` -> [Ps,Ps]."""
    labels = np.asarray(labels)
    Ps = len(labels)
    src, tgt = np.repeat(labels, Ps), np.tile(labels, Ps)
    return screw_fit(relative_trans(trans, src, tgt))["cost"].reshape(Ps, Ps)


def screw_cost(trans, connection):
    """gu:286-292 -> scalar, plus the reconstruction of gu:compute_screw_trans (:274-279)."""
    c = np.asarray(connection)
    T = np.asarray(trans).shape[0]
    f = screw_fit(relative_trans(trans, c[:, 0], c[:, 1]))
    recon = np.where((f["cost_p"] <= f["cost_r"])[None, :, None, None], f["recon_p"], f["recon_r"])
    return F32(f["cost"].mean(dtype=F32) / F32(T)), recon


def root_cost(trans):
    """gu:199-203."""
    e = np.asarray(trans, F32) - np.eye(4, dtype=F32)
    return (e * e).sum((2, 3), dtype=F32).mean(0, dtype=F32)


def part_fps(cano, seg, labels, num_fps=20):
    """gu:fps_sample_cano (:37-52) with the FPS of the reference's CUDA path (start index 0) -> idx into cano [Ps,F]."""
    cano, seg = np.asarray(cano, F32), np.asarray(seg)
    out = np.empty((len(labels), num_fps), np.int64)
    for k, p in enumerate(labels):
        members = np.nonzero(seg == p)[0]
        if len(members) < num_fps:
            raise ValueError("part id {} too small, only {} points".format(p, len(members)))
        out[k] = members[_o.fps(cano[members][None], num_fps, start=np.zeros(1, np.int64))[0]]
    return out


def part_pair_cost(cano, pred, fps_idx):
    """gu:compute_spatial_cost (:70-84) + gu:compute_joint_cost (:87-100) over all ordered part pairs:
    closest pair of the two FPS sets in the canonical frame (first minimum), and the squared distance of that pair
    summed over the predicted frames -> (cano_dist [Ps,Ps], pair [Ps,Ps,2], joint [Ps,Ps])."""
    cano, pred = np.asarray(cano, F32), np.asarray(pred, F32)
    Ps, Fn = fps_idx.shape
    pts = cano[fps_idx]                                         # [Ps,F,3]
    src = np.broadcast_to(pts[:, None], (Ps, Ps, Fn, 3)).reshape(-1, Fn, 3)
    tgt = np.broadcast_to(pts[None, :], (Ps, Ps, Fn, 3)).reshape(-1, Fn, 3)
    dmin, nn = _o.knn_points(np.ascontiguousarray(src), np.ascontiguousarray(tgt), K=1)
    dmin, nn = dmin[..., 0].reshape(Ps, Ps, Fn), nn[..., 0].reshape(Ps, Ps, Fn)
    s = dmin.argmin(-1)
    cd = np.take_along_axis(dmin, s[..., None], -1)[..., 0]
    t = np.take_along_axis(nn, s[..., None], -1)[..., 0]
    a = pred[:, fps_idx[np.arange(Ps)[:, None], s]]            # [T,Ps,Ps,3]: point s of part i
    b = pred[:, fps_idx[np.arange(Ps)[None, :], t]]            # point t of part j
    joint = ((a - b) ** 2).sum(-1, dtype=F32).sum(0, dtype=F32)
    return cd.astype(F32), np.stack([s, t], -1), joint


def mst(cost, labels=None, max_cost=None):
    """gu:295-324: repeated arg-min of cost + 1e10 * (same component) in fp32, first minimum in row-major order."""
    cost = np.asarray(cost, F32)
    n = cost.shape[0]
    conn = np.eye(n, dtype=np.int64)
    out = []
    for _ in range(n - 1):
        cur = cost + (conn * 1e10).astype(F32)
        k = int(cur.argmin())
        i, j = k // n, k % n
        if max_cost is not None and cur[i, j] > max_cost:
            break
        conn[i] = np.maximum(conn[i], conn[j])
        conn[conn[i] == 1] = conn[i]
        out.append([i, j] if labels is None else [int(labels[i]), int(labels[j])])
    return np.asarray(out, np.int64).reshape(-1, 2)


def contract_edges(edges, cost, merge_thr):
    """The networkx part of gu:344-385 -> (relabel {old: surviving}, remaining edges in M.edges order)."""
    import networkx as nx

    G = nx.DiGraph()
    for p in sorted({int(x) for e in edges for x in e}):
        G.add_node(p)
    for (a, b), c in zip(edges, cost):
        G.add_edge(int(a), int(b), cost=float(c))
    M = G.copy()
    relabel = {v: v for v in G.nodes}
    for node in list(nx.topological_sort(G)):
        if not M.has_node(node):
            continue
        for e in list(nx.edges(M, node)):
            if M.has_node(e[1]) and M.get_edge_data(e[0], e[1])["cost"] < merge_thr:
                M = nx.contracted_edge(M, e, self_loops=False)
                for k, v in relabel.items():
                    if v == e[1]:
                        relabel[k] = e[0]
    if not nx.is_weakly_connected(M):
        raise ValueError("New graph are not all connected.")
    if not nx.is_directed_acyclic_graph(M):
        raise ValueError("There are cycles in the link graph")
    return relabel, [[a, b] for a, b in M.edges]


def merge_graph(seg, connection, trans, merge_thr):
    """gu:327-385: contract tree edges whose relative motion stays within merge_thr of the identity."""
    seg = np.asarray(seg)
    c = np.asarray(connection)
    rel = relative_trans(trans, c[:, 0], c[:, 1])
    van = frobenius_cost(rel, np.broadcast_to(np.eye(4, dtype=F32), rel.shape)).mean(0, dtype=F32)
    relabel, remaining = contract_edges(c.tolist(), van.tolist(), merge_thr)
    lut = np.arange(max(int(seg.max()), max(relabel)) + 1)
    for k, v in relabel.items():
        lut[k] = v
    return lut[seg], np.asarray(remaining, np.int64).reshape(-1, 2)


def denoise_seg_label(seg, cano, min_num=10):
    """gu:115-123 (+ mu:knn_query :41-51 with k = 1): parts below min_num points take their nearest kept point's label."""
    seg, cano = np.asarray(seg).copy(), np.asarray(cano, F32)
    lab, cnt = np.unique(seg, return_counts=True)
    mask = np.isin(seg, lab[cnt < min_num])
    if mask.any():
        _, nn = _o.knn_points(cano[mask][None], cano[~mask][None], K=1)
        seg[mask] = seg[~mask][nn[0, :, 0]]
    return seg


def merging_wrapper(seg, trans, cano, merge_thr, n_it=2):
    """gu:388-416."""
    seg = np.asarray(seg).copy()
    pred = _o.compute_pc_transform(cano, trans, seg)
    for _ in range(n_it):
        labels = np.unique(seg)
        idx = part_fps(cano, seg, labels, 20)
        cd, _, joint = part_pair_cost(cano, pred, idx)
        cost = (cd + joint + F32(1e4) * np.eye(len(labels), dtype=F32)).astype(F32)
        seg, _ = merge_graph(seg, mst(cost, labels), trans, merge_thr)
        if len(np.unique(seg)) <= 1:
            break
    return seg


def mst_wrapper(seg, trans, cano, num_fps=20, cano_dist_thr=1e-2, joint_cost_weight=100.0):
    """gu:419-447."""
    labels = np.unique(seg)
    pred = _o.compute_pc_transform(cano, trans, seg)
    geo = geo_cost(trans, labels)
    cd, _, joint = part_pair_cost(cano, pred, part_fps(cano, seg, labels, num_fps))
    dist_cost = np.where(cd < cano_dist_thr, F32(0), F32(1e4)).astype(F32)
    cost = (dist_cost + geo + F32(joint_cost_weight) * joint).astype(F32)
    cost = cost + F32(1e4) * np.eye(len(labels), dtype=F32)
    return mst(cost, labels)


def extract_kinematic(seg, trans, connection):
    """ku:18-34: relabel the surviving parts 0..P-1."""
    labels = np.unique(seg)
    lut = {int(p): k for k, p in enumerate(labels)}
    new_seg = np.searchsorted(labels, seg)
    conn = np.asarray([[lut[int(a)], lut[int(b)]] for a, b in np.asarray(connection)], np.int64)
    return new_seg, np.asarray(trans, F32)[:, labels], conn


def build_graph(connection, trans):
    """ku:57-139 (revolute_only) + run_robot.py:119-121 -> dict(root, edges [(child,parent)], axis, moment, theta,
    paths_to_base, reverse_topo)."""
    import networkx as nx

    root = int(root_cost(trans).argmin())
    G0 = nx.from_edgelist(np.asarray(connection).tolist(), create_using=nx.Graph())
    paths = nx.shortest_path(G0, target=root)
    new_edges = []
    for p in G0.nodes:
        path = paths[p]
        for i in range(len(path) - 1):
            if (path[i], path[i + 1]) not in new_edges:
                new_edges.append((path[i], path[i + 1]))
    G = nx.from_edgelist(new_edges, create_using=nx.DiGraph())
    edges = list(G.edges())
    axis, moment, theta = [], [], []
    for child, parent in edges:
        l, m, th, _ = transform_to_screw(relative_trans(trans, [parent], [child])[:, 0])
        axis.append(l.mean(0, dtype=F32))
        moment.append(m.mean(0, dtype=F32))
        theta.append(th)
    return dict(root=root, edges=edges, axis=np.stack(axis), moment=np.stack(moment), theta=np.stack(theta, 1),
                paths_to_base=nx.shortest_path(G, target=root), reverse_topo=list(reversed(list(nx.topological_sort(G)))),
                nodes=list(G.nodes))


def group_temporal_err(pcs, seg):
    """mu:107-118: worst part's mean squared distance to its per-frame centroid."""
    pcs, seg = np.asarray(pcs, F32), np.asarray(seg)
    worst = F32(0)
    for p in np.unique(seg):
        part = pcs[:, seg == p].astype(np.float64)
        worst = max(worst, F32(((part - part.mean(1, keepdims=True)) ** 2).sum(2).mean()))
    return F32(worst)


def ass_err(pred, pcs):
    """mu:92-104: mean squared distance under the optimal one-to-one assignment of each frame (Euclidean cost)."""
    import torch

    pred, pcs = np.asarray(pred, F32), np.asarray(pcs, F32)
    cost = torch.cdist(torch.from_numpy(pred), torch.from_numpy(pcs)).numpy()
    tot = 0.0
    for b, (r, c) in enumerate(_o.linear_sum_assignment(cost)):
        tot += ((pred[b, r].astype(np.float64) - pcs[b, c]) ** 2).sum()
    return F32(tot / (pred.shape[0] * pred.shape[1]))
