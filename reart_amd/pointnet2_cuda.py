"""Drop-in for the live part of the reference's pybind module ``pointnet2_cuda``
(``networks/pointnet_lib/src/pointnet2_api.cpp:11-26``): the two wrappers the reference actually
calls (``furthest_point_sampling_wrapper`` from ``pointnet_lib/pointnet2_utils.py:28``,
``ball_query_wrapper`` from ``:262``), same argument order, caller-allocated int32 outputs,
return value 1.  The other eight wrappers are reachable only from ``pointnet2_modules.py``,
which nothing in the reference imports.  Of those, ``three_nn_wrapper`` has an exact
counterpart among the entry points that exist (the three nearest by direct-difference squared
distance, ascending scan: ``reart_knn_points_idx`` with K = 3) and is backed by it; the other seven
raise NotImplementedError (``three_interpolate_wrapper`` takes precomputed indices and weights in a
channel-major layout, which ``reart_three_interpolate`` -- fused with the search, point-major -- does not).
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


def _dead(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"pointnet2_cuda.{name} is dead code in the reference (never called); not built")
    fn.__name__ = name
    return fn


for _n in ("group_points_wrapper", "group_points_grad_wrapper", "gather_points_wrapper", "gather_points_grad_wrapper",
           "knn_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper"):
    globals()[_n] = _dead(_n)
