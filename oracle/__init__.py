"""oracle -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; ``reart_amd`` (the product) never does.  The arithmetic lives in the C files next
to this one (each function cites the reference file:line it restates); this module is the numpy
binding.  See oracle/oracle.h for the parity status (which parts are pinned by golden vectors
generated from the reference, and which are UNPINNED third-party contracts).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# REART_ORACLE_LIB: another build of the same sources (the sanitizer build, `make -C oracle asan`)
LIB_PATH = os.environ.get("REART_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")

_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        if hasattr(_lib, 'oracle_flow_loss'):
            _lib.oracle_flow_loss.restype = ctypes.c_double
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return lib().oracle_num_threads()


# ---- knn.c -------------------------------------------------------------------------------
def knn_points(p1, p2, lengths1=None, lengths2=None, K=1):
    """utils/chamfer.py:140-193 -> (dists f32 [N,P1,K], idx i64 [N,P1,K])."""
    p1, p2 = _f(p1), _f(p2)
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    l1 = None if lengths1 is None else np.ascontiguousarray(lengths1, dtype=np.int64)
    l2 = None if lengths2 is None else np.ascontiguousarray(lengths2, dtype=np.int64)
    dists = np.empty((N, P1, K), np.float32)
    idx = np.empty((N, P1, K), np.int64)
    lib().oracle_knn_points(_p(p1), _p(p2), _p(l1), _p(l2), N, P1, P2, D, K, _p(dists), _p(idx))
    return dists, idx


def knn_points_backward(p1, p2, idx, grad_dists, lengths1=None, lengths2=None):
    """utils/chamfer.py:195-209 -> (grad_p1, grad_p2)."""
    p1, p2, g = _f(p1), _f(p2), _f(grad_dists)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    K = idx.shape[2]
    l1 = None if lengths1 is None else np.ascontiguousarray(lengths1, dtype=np.int64)
    l2 = None if lengths2 is None else np.ascontiguousarray(lengths2, dtype=np.int64)
    g1, g2 = np.empty_like(p1), np.empty_like(p2)
    lib().oracle_knn_points_backward(_p(p1), _p(p2), _p(l1), _p(l2), _p(idx), _p(g), N, P1, P2, D, K, _p(g1), _p(g2))
    return g1, g2


def chamfer_bidir(x, y):
    """utils/chamfer.py:78-123 with bidirectional=True -> (d_xy, i_xy, d_yx, i_yx)."""
    d1, i1 = knn_points(x, y)
    d2, i2 = knn_points(y, x)
    return d1[..., 0], i1[..., 0], d2[..., 0], i2[..., 0]


def knn_cuda(ref, query, k, euclidean=True):
    """knn_cuda.KNN(k, transpose_mode=True)(ref, query) -> (dist [B,nq,k], idx [B,nq,k])."""
    ref, query = _f(ref), _f(query)
    B, nr, D = ref.shape
    nq = query.shape[1]
    dist = np.empty((B, nq, k), np.float32)
    idx = np.empty((B, nq, k), np.int64)
    lib().oracle_knn_cuda(_p(ref), _p(query), B, nr, nq, D, k, int(bool(euclidean)), _p(dist), _p(idx))
    return dist, idx


# ---- flow.c ------------------------------------------------------------------------------
def blend_anchor_motion(query, ref, ref_flow, k=3, euclidean=True):
    """utils/flow_utils.py:147-170 -> (flow [nq,3], mask bool [nq])."""
    query, ref, ref_flow = _f(query), _f(ref), _f(ref_flow)
    nq, nr = query.shape[0], ref.shape[0]
    flow = np.empty((nq, 3), np.float32)
    mask = np.empty((nq,), np.uint8)
    lib().oracle_blend_anchor_motion(_p(query), _p(ref), _p(ref_flow), nq, nr, k, int(bool(euclidean)),
                                     _p(flow), _p(mask))
    return flow, mask.astype(bool)


def flow_loss(gt, pred, mask=None, robust=False, smooth_weight=1e-2, want_grad=True):
    """networks/loss.py:10-21 -> (loss float, grad wrt pred or None)."""
    gt, pred = _f(gt), _f(pred)
    B, N, _ = pred.shape
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    grad = np.empty_like(pred) if want_grad else None
    loss = lib().oracle_flow_loss(_p(gt), _p(pred), _p(m), B, N, int(bool(robust)),
                                  ctypes.c_float(smooth_weight), _p(grad))
    return loss, grad


# ---- model.c -----------------------------------------------------------------------------
def rotation_6d_to_matrix(d6):
    d6 = _f(d6).reshape(-1, 6)
    R = np.empty((d6.shape[0], 3, 3), np.float32)
    lib().oracle_rotation_6d_to_matrix(_p(d6), d6.shape[0], _p(R))
    return R


def rotation_6d_backward(d6, gR):
    d6, gR = _f(d6).reshape(-1, 6), _f(gR).reshape(-1, 9)
    g = np.empty_like(d6)
    lib().oracle_rotation_6d_backward(_p(d6), _p(gR), d6.shape[0], _p(g))
    return g


def base_forward(cano, W1, b1, W2, p6d, pt, gumbel, tau):
    """networks/model.py:39-70 with injected noise ->
    dict(out [B,N,3], seg_part [N], trans_list [B,P,4,4], y_soft [N,P], hard_idx [N])."""
    cano, W1, b1, W2, p6d, pt, gumbel = map(_f, (cano, W1, b1, W2, p6d, pt, gumbel))
    N, (B, P), H = cano.shape[0], p6d.shape[:2], W1.shape[0]
    out = np.empty((B, N, 3), np.float32)
    seg = np.empty((N,), np.int64)
    trans = np.empty((B, P, 4, 4), np.float32)
    y = np.empty((N, P), np.float32)
    k = np.empty((N,), np.int32)
    lib().oracle_base_forward(_p(cano), N, P, B, _p(W1), _p(b1), _p(W2), H, _p(p6d), _p(pt), _p(gumbel),
                              ctypes.c_float(tau), _p(out), _p(seg), _p(trans), _p(y), _p(k))
    return dict(out=out, seg_part=seg, trans_list=trans, y_soft=y, hard_idx=k)


def base_backward(cano, W1, b1, W2, p6d, pt, y_soft, hard_idx, tau, G):
    cano, W1, b1, W2, p6d, pt, y_soft, G = map(_f, (cano, W1, b1, W2, p6d, pt, y_soft, G))
    hard_idx = np.ascontiguousarray(hard_idx, dtype=np.int32)
    N, (B, P), H = cano.shape[0], p6d.shape[:2], W1.shape[0]
    gW1, gb1, gW2 = np.empty_like(W1), np.empty_like(b1), np.empty_like(W2)
    g6d, gt = np.empty_like(p6d), np.empty_like(pt)
    lib().oracle_base_backward(_p(cano), N, P, B, _p(W1), _p(b1), _p(W2), H, _p(p6d), _p(pt), _p(y_soft),
                               _p(hard_idx), ctypes.c_float(tau), _p(G), _p(gW1), _p(gb1), _p(gW2), _p(g6d), _p(gt))
    return dict(gW1=gW1, gb1=gb1, gW2=gW2, g6d=g6d, gt=gt)


def compute_pc_transform(cano, pose, part):
    cano, pose = _f(cano), _f(pose)
    part = np.ascontiguousarray(part, dtype=np.int64)
    B, P = pose.shape[:2]
    out = np.empty((B, cano.shape[0], 3), np.float32)
    lib().oracle_compute_pc_transform(_p(cano), _p(pose), _p(part), cano.shape[0], P, B, _p(out))
    return out


def adam(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place torch.optim.Adam step on float32 arrays p, m, v."""
    assert p.dtype == np.float32 and p.flags.c_contiguous
    g = _f(g)
    lib().oracle_adam(_p(p), _p(g), _p(m), _p(v), p.size, step, ctypes.c_float(lr), ctypes.c_float(beta1),
                      ctypes.c_float(beta2), ctypes.c_float(eps))


# ---- pointnet.c --------------------------------------------------------------------------
def fps(xyz, npoint, start=None, cuda_mode=False):
    """networks/pointnet2_utils.py:74-99 (start injected) / sampling_gpu.cu:93-209 -> idx i64 [B,npoint]."""
    xyz = _f(xyz)
    B, N, _ = xyz.shape
    st = None if start is None else np.ascontiguousarray(start, dtype=np.int32)
    idx = np.empty((B, npoint), np.int64)
    lib().oracle_fps(_p(xyz), B, N, npoint, _p(st), int(bool(cuda_mode)), _p(idx))
    return idx


def ball_query(radius, nsample, xyz, new_xyz, cuda_mode=False, want_margin=False):
    """networks/pointnet2_utils.py:102-140 / ball_query_gpu.cu:9-45 -> idx i64 [B,S,nsample]
    (and per-row relative boundary margin min |d2-r2|/r2)."""
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = np.empty((B, S, nsample), np.int64)
    mg = np.empty((B, S), np.float32) if want_margin else None
    lib().oracle_ball_query(_p(xyz), _p(new_xyz), B, N, S, ctypes.c_double(radius), nsample, int(bool(cuda_mode)),
                            _p(idx), _p(mg))
    return (idx, mg) if want_margin else idx


def pn2_gather_points(points, idx):
    """networks/pointnet_lib/src/sampling_gpu.cu:8-24 (gather_points) and group_points_gpu.cu:39-54 (group_points: the same map
    over idx [B,npoints,nsample]): points f32 [B,C,N], idx int [B,...] -> out[b][c][...] = points[b][c][idx[b][...]].
    Parity unpinned: the CUDA kernels cannot run here and the reference holds no vectors for them (dead code there)."""
    points, idx = _f(points), np.asarray(idx)
    B, C, _ = points.shape
    out = np.empty((B, C) + idx.shape[1:], np.float32)
    for b in range(B):
        out[b] = points[b][:, idx[b]]
    return out


def pn2_gather_points_grad(grad_out, idx, N):
    """sampling_gpu.cu:46-63 / group_points_gpu.cu:8-21: grad_points[b][c][idx[b][...]] += grad_out[b][c][...] (the reference
    adds with float atomics in arrival order; here in float64, rounded once: compare with a tolerance)."""
    grad_out, idx = _f(grad_out), np.asarray(idx)
    B, C = grad_out.shape[:2]
    g = np.zeros((B, C, N), np.float64)
    for b in range(B):
        for c in range(C):
            np.add.at(g[b, c], idx[b].reshape(-1), grad_out[b, c].reshape(-1).astype(np.float64))
    return g.astype(np.float32)


def pn2_three_interpolate(points, idx, weight):
    """interpolate_gpu.cu:149-169: points f32 [B,C,M], idx int / weight f32 [B,N,3] -> out[b][c][n] = (w0 p[i0] + w1 p[i1]) +
    w2 p[i2] in fp32 without contraction, p = points[b][c][:]."""
    points, weight, idx = _f(points), _f(weight), np.asarray(idx)
    B, C, _ = points.shape
    N = idx.shape[1]
    out = np.empty((B, C, N), np.float32)
    for b in range(B):
        p = points[b][:, idx[b]]                       # [C,N,3]
        w = weight[b][None]                            # [1,N,3]
        out[b] = (w[..., 0] * p[..., 0] + w[..., 1] * p[..., 1]) + w[..., 2] * p[..., 2]
    return out


def pn2_three_interpolate_grad(grad_out, idx, weight, M):
    """interpolate_gpu.cu:192-214: grad_points[b][c][i_j] += grad_out[b][c][n] * w_j (products in fp32, the sums in float64 here)."""
    grad_out, weight, idx = _f(grad_out), _f(weight), np.asarray(idx)
    B, C, N = grad_out.shape
    g = np.zeros((B, C, M), np.float64)
    for b in range(B):
        for c in range(C):
            for j in range(3):
                np.add.at(g[b, c], idx[b][:, j], (grad_out[b, c] * weight[b][:, j]).astype(np.float64))
    return g.astype(np.float32)


def three_nn_expanded(xyz1, xyz2):
    """networks/pointnet2_utils.py:33-55 + :327-328: the 3 smallest matmul-expanded squared distances
    (stable ascending) of every xyz1 point in xyz2 -> (dists f32 [B,N,3], idx i64 [B,N,3])."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    d = np.empty((B, N, 3), np.float32)
    i = np.empty((B, N, 3), np.int64)
    lib().oracle_three_nn_expanded(_p(xyz1), _p(xyz2), B, N, S, _p(d), _p(i))
    return d, i


def three_interpolate(xyz1, xyz2, points2):
    """networks/pointnet2_utils.py:326-336: xyz1 [B,N,3], xyz2 [B,S,3], points2 [B,S,D] -> [B,N,D]."""
    xyz1, xyz2, points2 = _f(xyz1), _f(xyz2), _f(points2)
    B, N, _ = xyz1.shape
    S, D = points2.shape[1], points2.shape[2]
    out = np.empty((B, N, D), np.float32)
    lib().oracle_three_interpolate(_p(xyz1), _p(xyz2), _p(points2), B, N, S, D, _p(out))
    return out


# ---- screw.c -----------------------------------------------------------------------------
def se3_exp_map(log_transform):
    """screw_se3/geo_utils.py:147-222 -> [n,4,4] (pytorch3d row-vector form)."""
    lt = _f(log_transform)
    T = np.empty((lt.shape[0], 4, 4), np.float32)
    lib().oracle_se3_exp_map(_p(lt), lt.shape[0], _p(T))
    return T


def screw_to_transform(l, m, theta, d):
    """screw_se3/screw_utils.py:6-30 composed -> [n,4,4] column-vector transforms."""
    l, m, theta, d = _f(l), _f(m), _f(theta), _f(d)
    T = np.empty((l.shape[0], 4, 4), np.float32)
    lib().oracle_screw_to_transform(_p(l), _p(m), _p(theta), _p(d), l.shape[0], _p(T))
    return T


def fk(parent, edge_of_part, order, axis, moment, theta, distance=None):
    """utils/kinematic_utils.py:151-198 with the tree as arrays -> trans [B,P,4,4]."""
    parent = np.ascontiguousarray(parent, np.int32)
    eop = np.ascontiguousarray(edge_of_part, np.int32)
    order = np.ascontiguousarray(order, np.int32)
    axis, moment, theta = _f(axis), _f(moment), _f(theta)
    dist = None if distance is None else _f(distance)
    B, E = theta.shape
    P = parent.shape[0]
    T = np.empty((B, P, 4, 4), np.float32)
    lib().oracle_fk(_p(parent), _p(eop), _p(order), P, _p(axis), _p(moment), _p(theta), _p(dist), B, E, _p(T))
    return T


def linear_sum_assignment(cost):
    """The assignment loss's solver (reference run_robot.py:7,172-176; utils/model_utils.py:85-103): the
    reference calls scipy.optimize.linear_sum_assignment (requirements.txt:6, unpinned; 1.15.3 in this image)
    on every float32 `cdist` matrix.  No restatement: the reference's own third-party call IS the oracle here.
    cost [B,n,n] -> list of (row_ind, col_ind)."""
    from scipy.optimize import linear_sum_assignment as _lsa

    return [_lsa(np.asarray(c)) for c in np.asarray(cost)]


def parallel_lap(cost, nproc):
    """utils/model_utils.py:85-89 (`--use_nproc`, README.md:117,125): the reference ships the (T-1) matrices to a pool of
    `nproc = len(cost)` processes, one scipy solve each.  cost [B,n,n] -> list of (row_ind, col_ind)."""
    from multiprocessing import Pool

    from scipy.optimize import linear_sum_assignment as _lsa

    with Pool(processes=nproc) as pool:
        return pool.starmap_async(_lsa, zip(np.asarray(cost))).get()


def match_smnn(desc1, desc2, th=0.9):
    """Mutual second-nearest-neighbour ratio matching (reference utils/flow_utils.py:7-100: cdist -> topk(2) -> ratio
    test in both directions -> mutual filter, sorted by the desc1 index) -> (pairs [M,2] int64, ratio [M] = the larger
    of the two directions' ratios, margin = the closest any ratio comes to ``th``).  Distances in float64."""
    a, b = np.asarray(desc1, np.float64), np.asarray(desc2, np.float64)
    d2 = np.maximum((a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T, 0.0)
    dm = np.sqrt(d2)

    def top2(m):
        i1 = m.argmin(1)
        v1 = m[np.arange(m.shape[0]), i1]
        m2 = m.copy()
        m2[np.arange(m.shape[0]), i1] = np.inf
        return i1, v1 / m2.min(1)

    i12, r12 = top2(dm)
    i21, r21 = top2(dm.T)
    rows = np.nonzero((r12 <= th) & (r21[i12] <= th) & (i21[i12] == np.arange(dm.shape[0])))[0]
    pairs = np.stack([rows, i12[rows]], 1).astype(np.int64)
    margin = min(np.abs(r12 - th).min(), np.abs(r21 - th).min())
    return pairs, np.maximum(r12[rows], r21[i12[rows]]), float(margin)


def cdist(a, b):
    """Euclidean cost matrices of the assignment loss (reference run_robot.py:171 / utils/model_utils.py:93,
    `torch.cdist`), by the library-wide distance contract: sqrt(((dx*dx)+(dy*dy))+(dz*dz)) in fp32.  [B,n,3],[B,m,3] -> [B,n,m]."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    d = a[:, :, None, :] - b[:, None, :, :]
    sq = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return np.sqrt(sq, dtype=np.float32)
