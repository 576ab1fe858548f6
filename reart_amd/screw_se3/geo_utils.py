"""Hot-path subset of the reference's ``screw_se3/geo_utils.py``."""
import torch

from .. import _lib


def rotation_6d_to_matrix(d6):
    """Gram-Schmidt 6D -> rotation matrix, rows b1,b2,b3 (screw_se3/geo_utils.py:632-651).
    Forward-only entry point (inside BaseModel the conversion is fused with its backward)."""
    _lib.require_gpu(d6)
    flat = d6.reshape(-1, 6).contiguous().float()
    R = torch.empty((flat.shape[0], 3, 3), dtype=torch.float32, device=d6.device)
    rc = _lib.lib().reart_rotation_6d_to_matrix(_lib.ptr(flat), flat.shape[0], _lib.ptr(R), _lib.stream())
    _lib.check(rc, "reart_rotation_6d_to_matrix")
    return R.reshape(d6.shape[:-1] + (3, 3))


def matrix_to_rotation_6d(matrix):
    """First two rows, flattened (screw_se3/geo_utils.py:654-667)."""
    return matrix[..., :2, :].clone().reshape(matrix.shape[:-2] + (6,))


def inverse_transformation(trans_12):
    """[R|t] -> [R^T | -R^T t] for [N,4,4] or [4,4] (screw_se3/geo_utils.py:9-53)."""
    if not torch.is_tensor(trans_12):
        raise TypeError("Input type is not a torch.Tensor. Got {}".format(type(trans_12)))
    Rt = trans_12[..., :3, :3].transpose(-1, -2)
    out = torch.zeros_like(trans_12)
    out[..., :3, :3] = Rt
    out[..., :3, 3:4] = torch.matmul(-Rt, trans_12[..., :3, 3:4])
    out[..., 3, 3] = 1.0
    return out
