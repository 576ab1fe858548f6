"""Drop-in for the live part of the reference's pybind module ``pointnet2_cuda``
(``networks/pointnet_lib/src/pointnet2_api.cpp:11-26``): the two wrappers the reference actually
calls (``furthest_point_sampling_wrapper`` from ``pointnet_lib/pointnet2_utils.py:28``,
``ball_query_wrapper`` from ``:262``), same argument order, caller-allocated int32 outputs,
return value 1.  The other eight wrappers are reachable only from ``pointnet2_modules.py``,
which nothing in the reference imports; they raise NotImplementedError.
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


def _dead(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"pointnet2_cuda.{name} is dead code in the reference (never called); not built")
    fn.__name__ = name
    return fn


for _n in ("group_points_wrapper", "group_points_grad_wrapper", "gather_points_wrapper", "gather_points_grad_wrapper",
           "knn_wrapper", "three_nn_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper"):
    globals()[_n] = _dead(_n)
