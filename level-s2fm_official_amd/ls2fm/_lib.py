"""ctypes binding of libls2fm_hip.so (the C ABI declared in include/ls2fm.h).

The library is the product: there is NO fallback.  If it is missing or a call fails, a
RuntimeError is raised; CPU tensors are rejected (the CPU restatement lives in oracle/ and is test
infrastructure only).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p

import torch

MAX_LEVELS = 16
HIDDEN = 64
FEAT = 16
ABI_VERSION = 9
MAX_RENDER_POINTS = 1 << 23     # LS2FM_MAX_RENDER_POINTS

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LS2FM_LIB") or os.path.join(_HERE, "libls2fm_hip.so")


class GridDesc(Structure):
    _fields_ = [
        ("n_levels", c_int32),
        ("n_features", c_int32),
        ("scale", c_float * MAX_LEVELS),
        ("resolution", c_uint32 * MAX_LEVELS),
        ("size", c_uint32 * MAX_LEVELS),
        ("offset", c_uint32 * (MAX_LEVELS + 1)),
        ("hashed", c_uint32 * MAX_LEVELS),
    ]


class FieldDesc(Structure):
    _fields_ = [
        ("bound_min", c_float * 3),
        ("bound_max", c_float * 3),
        ("rescale", c_float),
        ("scale_mlp", c_float),
        ("inside", c_int32),
        ("bg_sdf", c_int32),
        ("bg_rad", c_float),
        ("bgcolor", c_float * 3),
        ("n_samples", c_int32),
        ("dual_field", c_int32),
    ]


class Linear(Structure):
    _fields_ = [("weight_v", c_void_p), ("weight_g", c_void_p), ("bias", c_void_p)]


class Params(Structure):
    _fields_ = [
        ("sdf_table", c_void_p),
        ("sdf_mlp", Linear * 2),
        ("beta", c_void_p),
        ("beta_speed", c_float),
        ("rad_table", c_void_p),
        ("geo_mlp", Linear * 2),
        ("rad_mlp", Linear * 3),
        ("dual_table", c_void_p),
    ]


class ParamGrads(Structure):
    _fields_ = [
        ("sdf_table", c_void_p),
        ("sdf_mlp", Linear * 2),
        ("beta", c_void_p),
        ("rad_table", c_void_p),
        ("geo_mlp", Linear * 2),
        ("rad_mlp", Linear * 3),
    ]


class LossSpec(Structure):
    """ls2fm_loss_spec: the loss head evaluated inside the render (forward epilogue / backward prologue)"""
    _fields_ = [("rgb_gt", c_void_p), ("depth_ref", c_void_p), ("mask_eik", c_void_p), ("mask_dc", c_void_p),
                ("mask_mse", c_void_p), ("weights", c_void_p), ("terms", c_void_p), ("sums", c_void_p),
                ("d_terms", c_void_p), ("d_total", c_void_p), ("d_depth_ref", c_void_p), ("flags", c_uint32),
                ("count_scale", c_uint32)]


LOSS_EIK_FROM_GT, LOSS_MSE_FROM_GT = 1, 2


class DepthBackward(Structure):
    _fields_ = [("points", c_void_p), ("trips", c_void_p), ("gate", c_void_p), ("k_max", c_int32), ("d_sdf", c_void_p),
                ("workspace", c_void_p)]


class RenderOpts(Structure):
    _fields_ = [("inference_only", c_int32), ("loss", POINTER(LossSpec)), ("n_level_groups", c_int32),
                ("group_events", c_void_p * 4), ("loss_inputs_ready", c_void_p), ("depth_grad_ready", c_void_p), ("depth_bwd", POINTER(DepthBackward))]


_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)        -- must list every symbol include/ls2fm.h declares
    "ls2fm_abi_version": (c_int32, []),
    "ls2fm_status_string": (c_char_p, [c_int32]),
    "ls2fm_async_error": (c_int32, [c_int32]),
    "ls2fm_ray_aabb_intersect": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P]),
    "ls2fm_grid_encode_fwd": (c_int32, [POINTER(GridDesc), _P, _P, c_int64, _P, _P, _P]),
    "ls2fm_grid_encode_bwd": (c_int32, [POINTER(GridDesc), _P, _P, _P, c_int64, _P, _P, _P]),
    "ls2fm_grid_encode_bwd_bwd": (c_int32, [POINTER(GridDesc), _P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "ls2fm_grid_indices": (c_int32, [POINTER(GridDesc), _P, c_int64, _P, _P]),
    "ls2fm_sdf_eval_workspace_bytes": (c_int64, []),
    "ls2fm_sdf_eval": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(Params), _P, c_int64, _P, _P, _P,
                                 _P, _P]),
    "ls2fm_sdf_volume": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(Params), c_int64, c_int64, c_int64,
                                   c_int32, _P, _P, _P, _P, _P]),
    "ls2fm_sdf_points_workspace_bytes": (c_int64, [POINTER(FieldDesc), POINTER(GridDesc), c_int64]),
    "ls2fm_sdf_points_bwd": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(Params), _P, c_int64, _P, _P, _P,
                                       POINTER(ParamGrads), _P, _P, _P]),
    "ls2fm_sdf_points_bwd_add": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(Params), _P, c_int64, _P, _P, _P,
                                           POINTER(ParamGrads), _P, _P, _P]),
    "ls2fm_render_workspace_bytes": (c_int64, [POINTER(FieldDesc), POINTER(GridDesc), c_int64]),
    "ls2fm_interleave_tables": (c_int32, [_P, _P, c_int64, _P, _P]),
    "ls2fm_render_fwd": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(GridDesc), POINTER(Params), _P, _P,
                                   c_int64, _P, _P, _P, _P, _P, _P, POINTER(RenderOpts), _P]),
    "ls2fm_render_bwd": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(GridDesc), POINTER(Params), _P, _P,
                                   c_int64, _P, _P, _P, _P, _P, POINTER(ParamGrads), _P, _P, _P, POINTER(RenderOpts), _P]),
    "ls2fm_loss_terms_from_sums": (c_int32, [_P, _P, _P, _P]),
    "ls2fm_sphere_trace": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(Params), _P, _P, c_int64,
                                     c_float, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ls2fm_sphere_trace_prepared": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(Params), _P, _P, c_int64,
                                              c_float, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ls2fm_sdf_prepare": (c_int32, [POINTER(GridDesc), POINTER(Params), _P, _P, _P]),
    "ls2fm_reproject_workspace_bytes": (c_int64, [c_int32]),
    "ls2fm_reproject_fwd": (c_int32, [_P, _P, _P, c_int32, POINTER(c_float), _P, _P, c_float, c_int64, _P, _P, _P, _P, _P]),
    "ls2fm_reproject_bwd": (c_int32, [_P, _P, _P, c_int32, POINTER(c_float), _P, _P, c_float, c_int64, _P, _P, _P, _P, _P]),
    "ls2fm_sdf_eval_prepared": (c_int32, [POINTER(FieldDesc), POINTER(GridDesc), POINTER(Params), _P, c_int64, _P, _P, _P]),
    "ls2fm_trace_depth_fwd": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, c_float, _P, c_float, c_float, _P, _P, _P, _P, _P, _P,
                                        _P]),
    "ls2fm_trace_depth_bwd": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P]),
    "ls2fm_loss_head_workspace_bytes": (c_int64, []),
    "ls2fm_loss_head_fwd": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P]),
    "ls2fm_loss_head_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P,
                                      _P]),
    "ls2fm_adam_step": (c_int32, [c_int32, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int64, _P]),
    "ls2fm_adam_step_scheduled": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, _P]),
    "ls2fm_adam_step_mirrored": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int64, _P]),
    "ls2fm_adam_step_multi": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_int64, _P]),
    "ls2fm_adam_sched_decay": (c_int32, [c_int32, _P, _P]),
    "ls2fm_camera_rays": (c_int32, [_P, _P, POINTER(c_float), _P, _P, c_int32, c_int32, _P, c_int32, c_int64, _P, _P, _P, _P]),
    "ls2fm_se3_exp_fwd": (c_int32, [_P, c_int32, _P, _P]),
    "ls2fm_se3_exp_bwd": (c_int32, [_P, _P, c_int32, _P, _P]),
    "ls2fm_tracing_term_fwd": (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, _P, _P]),
    "ls2fm_tracing_term_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "ls2fm_weighted_pair_fwd": (c_int32, [_P, _P, c_float, c_float, _P, _P]),
    "ls2fm_weighted_pair_bwd": (c_int32, [_P, c_float, c_float, _P, _P]),
    "ls2fm_ba_terms_fwd": (c_int32, [_P, _P, c_int64, _P, c_float, c_float, c_float, c_float, c_float, _P, _P, _P, _P]),
    "ls2fm_ba_terms_bwd": (c_int32, [_P, c_int64, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    "ls2fm_surface_pts_fwd": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P]),
    "ls2fm_surface_pts_bwd": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "ls2fm_match_term_fwd": (c_int32, [_P, _P, _P, _P, POINTER(c_float), c_int32, c_int64, _P, _P, _P, _P, _P]),
    "ls2fm_match_term_bwd": (c_int32, [_P, _P, _P, _P, POINTER(c_float), c_int32, c_int64, _P, _P, _P, _P, _P, _P]),
    "ls2fm_set_scatter_mode": (c_int32, [c_int32]),
    "ls2fm_get_scatter_mode": (c_int32, []),
    "ls2fm_profile_enable": (c_int32, [c_int32]),
    "ls2fm_profile_reset": (c_int32, []),
    "ls2fm_profile_count": (c_int32, []),
    "ls2fm_profile_name": (c_char_p, [c_int32]),
    "ls2fm_profile_get": (c_int32, [c_int32, POINTER(ctypes.c_double), POINTER(c_int64)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> ctypes.CDLL:
    """Load libls2fm_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C level-s2fm_official_amd/csrc`). "
            "There is no CPU or PyTorch fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ls2fm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libls2fm_hip.so ABI {lib.ls2fm_abi_version()} != python binding {ABI_VERSION}")
    _lib = lib
    return lib


def async_error(clear: bool = False) -> int:
    """sticky error word of the asynchronous parts of earlier calls (include/ls2fm.h: ls2fm_async_error); 0 = none"""
    return int(load().ls2fm_async_error(1 if clear else 0))


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().ls2fm_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")


def ptr(t) -> c_void_p:
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> c_void_p:
    """the caller's current HIP stream (raw handle: ~1 us; torch.cuda.current_stream() builds a Stream object, ~20 us)"""
    if _RAW_STREAM is not None:
        return c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(*tensors) -> None:
    """The kernels run on the GPU only; refuse anything else loudly (no CPU path in the product)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("ls2fm: this op runs on MI355X only (got a CPU tensor); there is no CPU fallback -- "
                               "the CPU restatement of the path is oracle/, used by the tests as the checker")
        if t.dtype not in (torch.float32, torch.int32, torch.int64, torch.uint8, torch.bool):
            raise RuntimeError(f"ls2fm: unsupported dtype {t.dtype}")


def cf(t: torch.Tensor) -> torch.Tensor:
    """contiguous float32 view/copy"""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
