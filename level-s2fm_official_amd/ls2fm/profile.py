"""Per-kernel device times from the library's opt-in HIP-event profiling (include/ls2fm.h, "Opt-in per-kernel
timing") and the roofline bookkeeping of bench.py.

Algorithmic work per launch (P = sample points of one batch, L = hash levels, 8 corners x 2 features
x 4 B = 64 B per level per point):
  ray_encode_*    : P * L * 64 B  gathered from the table (ray_encode_pair: both grids)  -> HBM roofline
  scatter (fill + accumulate) : P * L * 64 B per grid accumulated into the table gradient -> HBM roofline.  The two kernels
                    are two passes over ONE algorithmic transfer, so they are credited as a pair ("scatter_pair": bytes over
                    the SUM of their durations); neither is listed with a fraction of its own.
  shade_fwd       : P * 2 * MACs  (SDF MLP 35*64 + 64*17, normal W0^T 35*64 + 96, second field 35*64 + 64*17,
                    collapsed radiance 3*38)                                    -> f32 MFMA/VALU roofline
  shade_bwd       : P * 2 * MACs  (a, q, dE, r: 4 * 35*64; W1^T g 17*64; second field 2 * 35*64 + 16*64)
  wgrad_mlp_*     : 2 * P * sum(M*N) over the weight-gradient GEMMs of that MLP (f32 MFMA).  The hidden-layer
                    re-derivation the kernel performs instead of reading operands back from HBM is NOT counted as
                    algorithmic work.  (wgrad_dec: the decoder columns; reduce_finalize: the fixed-order sum of all three
                    producers' partials and the finalize tasks, one launch.)
While the profiler is enabled the library launches serially (no side-stream overlap), so every span is the duration of
that kernel alone and agrees with rocprofv3's per-kernel averages.
"""
from __future__ import annotations

import ctypes


def kernel_times(lib):
    """{name: (avg_us, launches, total_ms)} for every internal kernel that ran while profiling was enabled"""
    out = {}
    for i in range(lib.ls2fm_profile_count()):
        total = ctypes.c_double(0.0)
        launches = ctypes.c_int64(0)
        lib.ls2fm_profile_get(i, ctypes.byref(total), ctypes.byref(launches))
        if launches.value:
            out[lib.ls2fm_profile_name(i).decode()] = (total.value * 1e3 / launches.value, launches.value, total.value)
    return out


def algorithmic_work(n_points: int, dual: bool, n_levels: int = 16):
    """name -> (bound, work per launch, unit work): bytes for the table kernels, FLOPs for the dense ones"""
    table_bytes = n_points * n_levels * 64
    fwd_macs = 35 * 64 + 64 * 17 + 35 * 64 + 96 + 3 * (6 + 16 + (16 if dual else 0)) + (35 * 64 + 64 * 17 if dual else 0)
    bwd_macs = 4 * 35 * 64 + 17 * 64 + 96 + (2 * 35 * 64 + 16 * 64 if dual else 0)
    wgrad_mn = 64 * 36 + 64 * 35 + 17 * 65 + 64 + 3 * 39 + (64 * 36 + 17 * 65 if dual else 0)
    return {
        "ray_encode_sdf": ("hbm", table_bytes), "ray_encode_rad": ("hbm", table_bytes),
        "ray_encode_pair": ("hbm", 2 * table_bytes),             # dual field: both grids gathered by one launch
        # table-gradient scatter = scatter_fill (payload sort) + slab_accumulate, credited ONCE for the pair (_fractions)
        "scatter_pair": ("hbm", table_bytes * (2 if dual else 1)),
        "shade_fwd": ("mfma", 2 * fwd_macs * n_points), "shade_bwd": ("mfma", 2 * bwd_macs * n_points),
        "wgrad_mlp_sdf": ("mfma", 2 * (64 * 36 + 64 * 35 + 17 * 65 + 64) * n_points),
        "wgrad_mlp_geo": ("mfma", 2 * (64 * 36 + 17 * 65) * n_points),
        "_dual": dual,
    }


def _fused_wgrad(times, work):
    """round 5: shade_bwd contracts the MLPs' weight gradients itself (no wgrad_mlp launch in the render's backward) -- their
    GEMMs are then part of ITS algorithmic work"""
    if "shade_bwd" in times and "wgrad_mlp_sdf" not in times and "wgrad_mlp_geo" not in times:
        work = dict(work)
        work["shade_bwd"] = ("mfma", work["shade_bwd"][1] + work["wgrad_mlp_sdf"][1] + work["wgrad_mlp_geo"][1] * (1 if work.get("_dual") else 0))
    return work


def _fractions(times, work, hbm_peak_gbs, f32_peak_tflops):
    """algorithmic bytes (or FLOPs) / measured duration / peak, for every kernel with an algorithmic figure"""
    out = {}
    if "scatter_fill" in times and "slab_accumulate" in times:
        times = dict(times)
        pair_us = times["scatter_fill"][0] + times["slab_accumulate"][0]
        times["scatter_pair"] = (pair_us, times["scatter_fill"][1], times["scatter_fill"][2] + times["slab_accumulate"][2])
    for name, (avg_us, _, _) in times.items():
        if name not in work or name.startswith("_"):
            continue
        bound, amount = work[name]
        rate = amount / (avg_us * 1e-6)
        out[name] = {"bound": bound, "frac": round(rate / 1e9 / hbm_peak_gbs if bound == "hbm" else rate / 1e12 / f32_peak_tflops, 4)}
    return out


def dominant_kernel_roofline(lib, n_points: int, dual: bool, hbm_peak_gbs: float, f32_peak_tflops: float):
    """roofline object of bench.py's JSON line for the kernel with the largest share of device time"""
    times = kernel_times(lib)
    if not times:
        return None
    work = _fused_wgrad(times, algorithmic_work(n_points, dual))
    name = max(times, key=lambda k: times[k][2])          # a single kernel (the scatter's two passes are separate launches)
    avg_us, launches, total_ms = times[name]
    grand = sum(t[2] for t in times.values())
    bound, amount = work.get(name, ("hbm", 0))
    if name in ("scatter_fill", "slab_accumulate") and "scatter_fill" in times and "slab_accumulate" in times:
        # one of the two passes of ONE algorithmic transfer (the table-gradient scatter): reported as the pair -- its bytes over
        # the SUM of the two durations -- never one pass credited with the whole transfer
        bound, amount = work["scatter_pair"]
        name = "scatter_pair (scatter_fill + slab_accumulate)"
        avg_us = times["scatter_fill"][0] + times["slab_accumulate"][0]
        total_ms = times["scatter_fill"][2] + times["slab_accumulate"][2]
    if bound == "hbm":
        achieved, peak, unit = amount / (avg_us * 1e-6) / 1e9, hbm_peak_gbs, "GB/s"
    else:
        achieved, peak, unit = amount / (avg_us * 1e-6) / 1e12, f32_peak_tflops, "TFLOP/s"
    return {
        "kernel": name, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
        "traffic": None, "avg_launch_us": avg_us, "launches": launches, "share_of_device_time": total_ms / grand,
        "algorithmic_per_launch": amount,
        "all_kernels_avg_us": {k: round(v[0], 2) for k, v in sorted(times.items(), key=lambda kv: -kv[1][2])},
        "all_kernels_frac_of_peak": _fractions(times, work, hbm_peak_gbs, f32_peak_tflops),
        # FLOPs the dense kernels of one step actually EXECUTE (collapsed radiance decoder, fused weight gradients): what a whole-step
        # MFMA fraction may be credited with (SURVEY 8d's 115 kFLOP per sample counts the un-collapsed decoder, which nothing runs)
        "useful_flops_per_step": float(sum(work[k][1] for k in ("shade_fwd", "shade_bwd", "wgrad_mlp_sdf", "wgrad_mlp_geo")
                                           if k in times and k in work)),
    }
