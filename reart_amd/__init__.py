"""reart_amd -- MI355X (gfx950) implementation of reart's per-iteration point-cloud hot path.

The package is a thin host-side mirror of the reference's Python interfaces
(``utils/chamfer.py``, ``knn_cuda.KNN``, ``networks/model.py``, ``networks/loss.py``,
``networks/pointnet_lib``) over the C ABI of ``libreart_hip.so`` (``include/reart_hip.h``).
There is no CPU fallback: every operator raises if the HIP library or a GPU is missing.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
