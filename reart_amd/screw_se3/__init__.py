"""Hot-path subset of the reference's ``screw_se3`` package (HIP-backed)."""
from .geo_utils import rotation_6d_to_matrix, matrix_to_rotation_6d, inverse_transformation  # noqa: F401
