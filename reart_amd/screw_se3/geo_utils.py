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
