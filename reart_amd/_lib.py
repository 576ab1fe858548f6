"""ctypes binding of libreart_hip.so (C ABI declared in include/reart_hip.h).

PyTorch is used for device memory and streams only: tensors are passed as raw device
pointers (``tensor.data_ptr()``) together with torch's current HIP stream.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# REART_LIB overrides the library path (kernel A/B experiments); the default is the in-tree build
LIB_PATH = os.environ.get("REART_LIB") or os.path.join(_HERE, "csrc", "libreart_hip.so")

c_int, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
P = c_void_p  # every device pointer

# name -> (restype, argtypes); mirrors include/reart_hip.h one to one
PROTOTYPES = {
    "reart_version": (c_int, []),
    "reart_device_count": (c_int, []),
    "reart_status_string": (ctypes.c_char_p, [c_int]),
    "reart_knn_points_workspace_bytes": (c_size_t, [c_int] * 4),
    "reart_knn_points_idx": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "reart_knn_points_backward_workspace_bytes": (c_size_t, [c_int] * 4),
    "reart_knn_points_backward": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "reart_chamfer_bidir_workspace_bytes": (c_size_t, [c_int] * 2),
    "reart_chamfer_bidir": (c_int, [P, P, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_knn_cuda": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "reart_blend_anchor_motion_workspace_bytes": (c_size_t, [c_int] * 3),
    "reart_blend_anchor_motion": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "reart_kin_post_workspace_bytes": (c_size_t, [c_int] * 4),
    "reart_kin_post": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, c_int, c_float, P, P, P, c_int, c_int, c_int, c_float, c_int, c_float,
                               P, P, P, P, c_size_t, P]),
    "reart_blend_anchor_motion_batch_workspace_bytes": (c_size_t, [c_int] * 4),
    "reart_blend_anchor_motion_batch": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "reart_flow_loss_workspace_bytes": (c_size_t, []),
    "reart_flow_loss": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P, P, P, c_size_t, P]),
    "reart_base_forward": (c_int, [P, c_int, c_int, c_int, P, P, P, c_int, P, P, P, c_float, P, P, P, P, P, P, P]),
    "reart_gumbel_noise": (c_int, [ctypes.c_uint64, ctypes.c_int64, c_int, c_int, P, P]),
    "reart_base_backward_workspace_bytes": (c_size_t, [c_int] * 4),
    "reart_base_backward": (c_int, [P, c_int, c_int, c_int, P, P, P, c_int, P, P, P, P, P, c_float, P,
                                    P, P, P, P, P, P, c_size_t, P]),
    "reart_compute_pc_transform": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "reart_rotation_6d_to_matrix": (c_int, [P, c_int, P, P]),
    "reart_adam_step": (c_int, [P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float, P]),
    "reart_adam_step_multi": (c_int, [c_int, P, P, P, P, P, P, c_int, c_float, c_float, c_float, P]),
    "reart_fps": (c_int, [P, c_int, c_int, c_int, P, c_int, P, P, P]),
    "reart_ball_query": (c_int, [P, P, c_int, c_int, c_int, ctypes.c_double, c_int, c_int, P, P, P]),
    "reart_pn2_gather_points": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "reart_pn2_gather_points_grad": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "reart_pn2_three_interpolate": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "reart_pn2_three_interpolate_grad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "reart_fk_forward": (c_int, [P, P, P, c_int, P, P, P, P, c_int, c_int, P, P]),
    "reart_fk_backward_workspace_bytes": (c_size_t, [c_int] * 3),
    "reart_fk_backward": (c_int, [P, P, P, c_int, P, P, P, c_int, P, P, P, P, c_int, c_int, P, P, P, P, P, P,
                                  c_size_t, P]),
    "reart_mlp_layer": (c_int, [P, c_int, P, c_int, c_int, c_int, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int,
                                c_int, c_int, P, c_int, c_int, P]),
    "reart_mlp_chain3": (c_int, [P, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, c_int, P, P, c_int, c_int, P, c_int, c_int, P]),
    "reart_mlp_chain3_wide_workspace_bytes": (c_size_t, [c_int] * 4),
    "reart_mlp_chain3_wide": (c_int, [P, c_int, c_int, c_int, P, c_int, P, P, P, P, c_int, P, P, c_int, P, P, c_int, c_int, P, c_int,
                                      c_int, P, c_size_t, P]),
    "reart_three_nn": (c_int, [P, P, c_int, c_int, c_int, P, P, P]),
    "reart_three_interpolate_workspace_bytes": (c_size_t, [c_int] * 3),
    "reart_three_interpolate": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_size_t, P]),
    "reart_grid_knn_workspace_bytes": (c_size_t, [c_int] * 2),
    "reart_grid_knn": (c_int, [P, P, c_int, c_int, P, c_int, c_int, P, P, P, c_size_t, P]),
    "reart_knn_points_warm_workspace_bytes": (c_size_t, [c_int] * 4),
    "reart_knn_points_idx_warm": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "reart_lap_workspace_bytes": (c_size_t, [c_int] * 2),
    "reart_lap_auction": (c_int, [P, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_lap_auction_warm": (c_int, [P, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_lap_auction_points": (c_int, [P, P, P, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_lap_race_workspace_bytes": (c_size_t, [c_int] * 3),
    "reart_lap_auction_race": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "reart_lap_auction_race_warm": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_size_t, P]),
    "reart_lap_resolve": (c_int, [P, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_lap_resolve_points": (c_int, [P, P, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_lap_resolve_points_race": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_lap_resolve_points_mw": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_lap_mc_workspace_bytes": (c_size_t, [c_int] * 3),
    "reart_lap_resolve_points_mc": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "reart_gather_points": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "reart_publish_words": (c_int, [P, c_int, P, c_int, P, c_int, P, P]),
    "reart_assign_pairs": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "reart_lap_ties": (c_int, [P, P, c_int, c_int, P, P, P, P, P, c_int, P]),
    "reart_lap_resolve_points_mc_ties": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int, P, c_size_t, P]),
    "reart_lap_step_floor": (c_int, [c_int, c_int, c_int, P, c_size_t, ctypes.POINTER(ctypes.c_double), P]),
    "reart_relax_step_floor": (c_int, [ctypes.POINTER(c_int), c_int, c_int, P, c_size_t, P]),
    "reart_cdist": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "reart_match_smnn_workspace_bytes": (c_size_t, [c_int] * 3),
    "reart_match_smnn": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, P, P, P, c_size_t, P]),
    "reart_screw_fit_workspace_bytes": (c_size_t, [c_int] * 2),
    "reart_screw_fit": (c_int, [P, c_int, c_int, P, c_int, c_int, P, P, P, P, P, P, P, c_size_t, P]),
    "reart_part_fps": (c_int, [P, P, c_int, P, c_int, c_int, c_int, P, P, P]),
    "reart_part_pair_cost": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P]),
    "reart_group_temporal_err": (c_int, [P, c_int, c_int, P, P, c_int, P, P, P]),
    # struct-taking entry points: full prototypes are set in reart_amd/relax.py
    "reart_relax_workspace_bytes": (c_size_t, None),
    "reart_relax_prepare": (c_int, None),
    "reart_relax_step": (c_int, None),
    "reart_relax_step_batch": (c_int, None),
    "reart_relax_forward": (c_int, None),
    "reart_relax_step_timed": (c_int, None),
    "reart_relax_profile": (c_int, None),
}

_lib = None
_lock = threading.Lock()


class ReartHipError(RuntimeError):
    """A libreart_hip.so entry point returned a negative reart_status."""


def lib():
    """Load libreart_hip.so (once).  Fails loudly: there is no fallback implementation."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} is missing: build it with `make -C reart_amd/csrc -j8` "
                        "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                        "reart_amd has no CPU / PyTorch fallback."
                    )
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in PROTOTYPES.items():
                    fn = getattr(handle, name)  # AttributeError if the .so is stale
                    fn.restype = res
                    if args is not None:
                        fn.argtypes = args
                _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().reart_status_string(rc).decode()
        raise ReartHipError(f"{what} failed: {msg} (status {rc})")


def require_gpu(*tensors):
    """Every operand must live on a HIP device; reart_amd never computes on the host."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "reart_amd operators run on an AMD GPU only (tensor is on "
                f"{t.device}); there is no CPU fallback."
            )


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_ws_cache = {}


def workspace(nbytes, device):
    """Scratch buffer reused across calls on the same (device, stream): calls on one
    stream are ordered, so a shared buffer is safe; the library itself never allocates."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
