"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/reart_hip.h declares, and the host mirror raises the reference's errors.  No compute."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "reart_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(reart_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from reart_amd import _lib

    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/reart_hip.h but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype in reart_amd/_lib.py"
    assert set(_lib.PROTOTYPES) == set(names)
    assert L.reart_version() >= 100
    assert L.reart_status_string(-2) == b"unsupported configuration"


def test_no_cpu_fallback():
    from reart_amd.utils.chamfer import ChamferDistance, knn_points
    from reart_amd.knn_cuda import KNN

    a = torch.zeros(1, 8, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        knn_points(a, a)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ChamferDistance()(a, a)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        KNN(k=1, transpose_mode=True)(a, a)


def test_chamfer_argument_errors_match_reference():
    """Same exception types as utils/chamfer.py:34-76,261-264."""
    from reart_amd.utils.chamfer import ChamferDistance, knn_points

    cd = ChamferDistance()
    a = torch.zeros(2, 8, 3)
    with pytest.raises(TypeError):
        cd([1, 2], a)
    with pytest.raises(ValueError, match="same batchsize"):
        cd(a, torch.zeros(3, 8, 3))
    with pytest.raises(ValueError, match="same dimensionality"):
        cd(a, torch.zeros(2, 8, 2))
    with pytest.raises(ValueError, match="Reduction"):
        cd(a, a, reduction="max")
    with pytest.raises(ValueError, match="same batch dimension"):
        knn_points(a, torch.zeros(3, 8, 3))
    with pytest.raises(ValueError, match="same point dimension"):
        knn_points(a, torch.zeros(2, 8, 4))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under reart_amd/ may reference it."""
    pkg = os.path.join(ROOT, "reart_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
