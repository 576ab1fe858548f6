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


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument checks come before any device work: status codes, never exit() (SURVEY.md 8b error convention)."""
    from reart_amd import _lib

    L = _lib.lib()
    bad = -1   # REART_ERR_INVALID_ARG
    assert L.reart_screw_fit(None, 3, 2, None, 4, 0, None, None, None, None, None, None, None, 0, None) == bad
    assert L.reart_part_fps(None, None, 10, None, 2, 20, 0, None, None, None) == bad
    assert L.reart_part_pair_cost(None, None, 0, 3, 20, None, None, None, None) == bad
    assert L.reart_group_temporal_err(None, 2, 10, None, None, 3, None, None, None) == bad
    assert L.reart_cdist(None, None, 1, 4, 4, None, None) == bad
    assert L.reart_cdist(None, None, 1, 70000, 4, None, None) == bad
    assert L.reart_lap_auction(None, 1, 5000, None, None, None, None, None, 0, None) == bad      # n above 4096
    assert L.reart_lap_workspace_bytes(3, 4096) > L.reart_lap_workspace_bytes(3, 2048) > 0
    assert L.reart_lap_workspace_bytes(3, 4097) == 0
    assert L.reart_screw_fit_workspace_bytes(19, 400) == 400 * 6 * 4
    # the measurement aid of the headline's roofline: shapes are validated before anything is launched
    import ctypes
    shape = (ctypes.c_int * 5)(128, 320, 4096, 1, 6)
    assert L.reart_relax_step_floor(None, 1, 1, 1, 1 << 20, None) == bad
    assert L.reart_relax_step_floor(shape, 1, 1, None, 1 << 20, None) == bad
    assert L.reart_relax_step_floor(shape, 9, 1, 1, 1 << 20, None) == bad                    # at most eight launches per iteration
    assert L.reart_relax_step_floor(shape, 1, 1, 1, 1024, None) == bad                       # workspace below 64 KB
    shape[1] = 2048
    assert L.reart_relax_step_floor(shape, 1, 1, 1, 1 << 20, None) == bad                    # block size above 1024
    # empty problems are fine
    assert L.reart_part_fps(1, 1, 10, 1, 0, 20, 0, 1, 1, None) == 0
    assert L.reart_cdist(None, None, 0, 4, 4, None, None) == 0


def test_structure_wrappers_have_no_cpu_fallback():
    from reart_amd.utils import graph_utils as gu
    from reart_amd.utils.lap import cdist
    from reart_amd.utils.model_utils import compute_ass_err, compute_group_temporal_err

    t = torch.eye(4).repeat(2, 3, 1, 1)
    for call in (lambda: gu.screw_fit(t), lambda: gu.fps_sample_cano(torch.zeros(8, 3), torch.zeros(8, dtype=torch.long), torch.zeros(1, dtype=torch.long)),
                 lambda: gu.compute_spatial_cost(torch.zeros(2, 4, 3)), lambda: cdist(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3)),
                 lambda: compute_ass_err(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3)),
                 lambda: compute_group_temporal_err(torch.zeros(2, 4, 3), torch.zeros(4, dtype=torch.long))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


def test_relax_config_struct_matches_the_header():
    """The ctypes mirror of reart_relax_config lists the header's fields in the header's order with the header's types
    (a silent mismatch would shift every tuning field)."""
    import ctypes
    import re

    from reart_amd.relax import RelaxConfig

    hdr = open(os.path.join(ROOT, "include", "reart_hip.h")).read()
    body = hdr[hdr.index("typedef struct reart_relax_config {"):hdr.index("} reart_relax_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for typ, names in re.findall(r"\b(int|float|uint64_t)\s+([a-zA-Z_0-9,\s]+);", body):
        for nm in names.split(","):
            fields.append((nm.strip(), typ))
    ctype = {"int": ctypes.c_int, "float": ctypes.c_float, "uint64_t": ctypes.c_uint64}
    assert [(n, ctype[t]) for n, t in fields] == list(RelaxConfig._fields_)
