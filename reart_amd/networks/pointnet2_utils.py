"""Host-side mirror of the sampling / grouping functions of the reference's
``networks/pointnet2_utils.py`` over the HIP kernels (``reart_fps``, ``reart_ball_query``).

The reference picks its sampling rules with a module-level switch (``CUDA``, networks/pointnet2_utils.py:7-12:
true whenever a GPU is present) -- mirrored here as ``CUDA = True``, because this package only runs on a GPU:
  * ``CUDA`` rules (default): the vendored CUDA kernels -- FPS starts at index 0 with the block-tree tie rule
    (sampling_gpu.cu:93-209), ball query keeps ``d2 < r*r`` on coordinate differences and pads with the first hit
    (ball_query_gpu.cu:9-45);
  * CPU-fallback rules (``cuda_mode=False`` per call, or ``pointnet2_utils.CUDA = False``): FPS starts at a
    ``torch.randint`` draw, arg-max = first maximum (:88-99); ball query keeps ``d2 <= r^2`` on the matmul-expanded
    ``square_distance`` and pads with the nearest point (:102-140) -- BASELINE's "reference CPU/PyTorch path", the
    rules the CPU-generated golden vectors follow.
"""
import torch

from .. import _lib

CUDA = True   # networks/pointnet2_utils.py:7-12


def _rules(cuda_mode):
    return CUDA if cuda_mode is None else bool(cuda_mode)


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] -> [B,S,(K,)C] (networks/pointnet2_utils.py:54-71)."""
    B = points.shape[0]
    flat = idx.reshape(B, -1)
    out = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1]))
    return out.reshape(*idx.shape, points.shape[-1])


def farthest_point_sample(xyz, npoint, start=None, cuda_mode=None):
    """xyz [B,N,3] -> int64 [B,npoint] (networks/pointnet2_utils.py:74-99).

    ``start`` [B]: first index of every cloud.  The reference's CPU fallback draws it with
    ``torch.randint`` from the global generator (:90) -- reproduced here when ``start`` is None and
    ``cuda_mode`` is False; its CUDA kernel always starts at 0 (sampling_gpu.cu:113)."""
    _lib.require_gpu(xyz)
    cuda_mode = _rules(cuda_mode)
    xyz = xyz.contiguous().float()
    B, N, _ = xyz.shape
    if start is None and not cuda_mode:
        start = torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device)
    st = None if start is None else start.to(device=xyz.device, dtype=torch.int32).contiguous()
    idx = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
    rc = _lib.lib().reart_fps(_lib.ptr(xyz), B, N, npoint, _lib.ptr(st), int(bool(cuda_mode)), None, _lib.ptr(idx),
                              _lib.stream())
    _lib.check(rc, "reart_fps")
    return idx


def query_ball_point(radius, nsample, xyz, new_xyz, cuda_mode=None):
    """xyz [B,N,3], new_xyz [B,S,3] -> int64 [B,S,nsample] (networks/pointnet2_utils.py:102-140)."""
    _lib.require_gpu(xyz, new_xyz)
    cuda_mode = _rules(cuda_mode)
    xyz, new_xyz = xyz.contiguous().float(), new_xyz.contiguous().float()
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    rc = _lib.lib().reart_ball_query(_lib.ptr(xyz), _lib.ptr(new_xyz), B, N, S, float(radius), nsample,
                                     int(bool(cuda_mode)), None, _lib.ptr(idx), _lib.stream())
    _lib.check(rc, "reart_ball_query")
    return idx
