"""Drop-in for the reference's pybind module ``pointnet2_cuda`` (``networks/pointnet_lib/src/pointnet2_api.cpp:11-26``), all
ten wrappers, same argument order, caller-allocated outputs.  The reference itself calls two of them
(``furthest_point_sampling_wrapper`` from ``pointnet_lib/pointnet2_utils.py:29``, ``ball_query_wrapper`` from ``:263``); the other
eight are reachable only from ``pointnet2_modules.py``, which nothing in the reference imports -- they are here so
that the module is whole: ``three_nn_wrapper`` / ``knn_wrapper`` on the K-nearest search that exists (direct-difference squared
distance, ascending scan, ties keep the lower index: ``reart_knn_points_idx``, K <= 16), the channel-major gather / group /
interpolate operators and their backward forms on ``reart_pn2_*`` (csrc/pointnet.hip).  Their parity is against
restatements of the CUDA kernels in ``oracle/`` (the reference's kernels cannot run here and hold no golden vectors: parity
unpinned for these eight).
"""
from . import _lib


def furthest_point_sampling_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    """points f32 [B,N,3], temp f32 [B,N] (unused: distances live in registers), idx i32 [B,M]."""
    _lib.require_gpu(points_tensor, idx_tensor)
    rc = _lib.lib().reart_fps(_lib.ptr(points_tensor), b, n, m, None, 1, _lib.ptr(idx_tensor), None, _lib.stream())
    _lib.check(rc, "reart_fps")
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz_tensor, xyz_tensor, idx_tensor):
    """new_xyz f32 [B,M,3], xyz f32 [B,N,3], idx i32 [B,M,nsample]; CUDA-kernel semantics."""
    _lib.require_gpu(new_xyz_tensor, xyz_tensor, idx_tensor)
    if not new_xyz_tensor.is_contiguous() or not xyz_tensor.is_contiguous():
        raise RuntimeError("tensors must be contiguous")  # CHECK_CONTIGUOUS, ball_query.cpp:12
    rc = _lib.lib().reart_ball_query(_lib.ptr(xyz_tensor), _lib.ptr(new_xyz_tensor), b, n, m, float(radius), nsample,
                                     1, _lib.ptr(idx_tensor), None, _lib.stream())
    _lib.check(rc, "reart_ball_query")
    return 1


def three_nn_wrapper(b, n, m, unknown_tensor, known_tensor, dist2_tensor, idx_tensor):
    """interpolate.cpp:15-26 / interpolate_gpu.cu:81-131: unknown f32 [B,N,3], known f32 [B,M,3] -> dist2 f32 [B,N,3],
    idx i32 [B,N,3]: the three nearest `known` points of every `unknown` point by squared distance
    (ux-x)^2 + (uy-y)^2 + (uz-z)^2, scanning k ascending with strict `<` (ties keep the lower index)."""
    from .chamferdist_C import knn_points_idx

    _lib.require_gpu(unknown_tensor, known_tensor, dist2_tensor, idx_tensor)
    idx, dists = knn_points_idx(unknown_tensor.reshape(b, n, 3), known_tensor.reshape(b, m, 3), None, None, 3)
    dist2_tensor.reshape(b, n, 3).copy_(dists)
    idx_tensor.reshape(b, n, 3).copy_(idx)
    return None


def knn_wrapper(b, n, m, k, unknown_tensor, known_tensor, dist2_tensor, idx_tensor):
    """interpolate.cpp:27-37 / interpolate_gpu.cu:9-58: unknown f32 [B,N,3], known f32 [B,M,3] -> dist2 f32 [B,N,k], idx i32
    [B,N,k], the k nearest by the same squared distance, insertion-sorted ascending with strict `<` (ties keep the lower
    index).  k <= 16 here (the kernel's register list; the reference's bound is its 200-entry local array)."""
    from .chamferdist_C import knn_points_idx

    if k > 16:
        raise NotImplementedError("pointnet2_cuda.knn_wrapper: k <= 16 (REART_MAX_K)")
    _lib.require_gpu(unknown_tensor, known_tensor, dist2_tensor, idx_tensor)
    idx, dists = knn_points_idx(unknown_tensor.reshape(b, n, 3), known_tensor.reshape(b, m, 3), None, None, k)
    dist2_tensor.reshape(b, n, k).copy_(dists)
    idx_tensor.reshape(b, n, k).copy_(idx)
    return None


def _contig(*ts):
    for t_ in ts:
        if not t_.is_contiguous():
            raise RuntimeError("tensors must be contiguous")       # CHECK_CONTIGUOUS in every wrapper of the reference


def gather_points_wrapper(b, c, n, npoints, points_tensor, idx_tensor, out_tensor):
    """sampling.cpp:11-22 / sampling_gpu.cu:8-24: points f32 [B,C,N], idx i32 [B,npoints] -> out f32 [B,C,npoints]."""
    _lib.require_gpu(points_tensor, idx_tensor, out_tensor)
    _contig(points_tensor, idx_tensor, out_tensor)
    _lib.check(_lib.lib().reart_pn2_gather_points(_lib.ptr(points_tensor), _lib.ptr(idx_tensor), b, c, n, npoints, _lib.ptr(out_tensor),
                                                  _lib.stream()), "reart_pn2_gather_points")
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out_tensor, idx_tensor, grad_points_tensor):
    """sampling.cpp:24-35 / sampling_gpu.cu:46-63: grad_out f32 [B,C,npoints] accumulated into grad_points f32 [B,C,N] (zeroed
    by the caller, pointnet_lib/pointnet2_utils.py:70)."""
    _lib.require_gpu(grad_out_tensor, idx_tensor, grad_points_tensor)
    _contig(grad_out_tensor, idx_tensor, grad_points_tensor)
    _lib.check(_lib.lib().reart_pn2_gather_points_grad(_lib.ptr(grad_out_tensor), _lib.ptr(idx_tensor), b, c, n, npoints,
                                                       _lib.ptr(grad_points_tensor), _lib.stream()), "reart_pn2_gather_points_grad")
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points_tensor, idx_tensor, out_tensor):
    """group_points.cpp:26-38 / group_points_gpu.cu:39-54: points f32 [B,C,N], idx i32 [B,npoints,nsample] -> out f32
    [B,C,npoints,nsample]: the gather above over npoints * nsample indices per batch."""
    _lib.require_gpu(points_tensor, idx_tensor, out_tensor)
    _contig(points_tensor, idx_tensor, out_tensor)
    _lib.check(_lib.lib().reart_pn2_gather_points(_lib.ptr(points_tensor), _lib.ptr(idx_tensor), b, c, n, npoints * nsample,
                                                  _lib.ptr(out_tensor), _lib.stream()), "reart_pn2_gather_points")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out_tensor, idx_tensor, grad_points_tensor):
    """group_points.cpp:12-23 / group_points_gpu.cu:8-21: grad_out f32 [B,C,npoints,nsample] accumulated into grad_points."""
    _lib.require_gpu(grad_out_tensor, idx_tensor, grad_points_tensor)
    _contig(grad_out_tensor, idx_tensor, grad_points_tensor)
    _lib.check(_lib.lib().reart_pn2_gather_points_grad(_lib.ptr(grad_out_tensor), _lib.ptr(idx_tensor), b, c, n, npoints * nsample,
                                                       _lib.ptr(grad_points_tensor), _lib.stream()), "reart_pn2_gather_points_grad")
    return 1


def three_interpolate_wrapper(b, c, m, n, points_tensor, idx_tensor, weight_tensor, out_tensor):
    """interpolate.cpp:40-54 / interpolate_gpu.cu:149-169: points f32 [B,C,M], idx i32 / weight f32 [B,N,3] -> out f32 [B,C,N]."""
    _lib.require_gpu(points_tensor, idx_tensor, weight_tensor, out_tensor)
    _contig(points_tensor, idx_tensor, weight_tensor, out_tensor)
    _lib.check(_lib.lib().reart_pn2_three_interpolate(_lib.ptr(points_tensor), _lib.ptr(idx_tensor), _lib.ptr(weight_tensor), b, c, m, n,
                                                      _lib.ptr(out_tensor), _lib.stream()), "reart_pn2_three_interpolate")
    return None


def three_interpolate_grad_wrapper(b, c, n, m, grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor):
    """interpolate.cpp:56-70 / interpolate_gpu.cu:192-214: grad_out f32 [B,C,N] accumulated into grad_points f32 [B,C,M]."""
    _lib.require_gpu(grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor)
    _contig(grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor)
    _lib.check(_lib.lib().reart_pn2_three_interpolate_grad(_lib.ptr(grad_out_tensor), _lib.ptr(idx_tensor), _lib.ptr(weight_tensor), b, c, n,
                                                           m, _lib.ptr(grad_points_tensor), _lib.stream()), "reart_pn2_three_interpolate_grad")
    return None
