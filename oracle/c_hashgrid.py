"""ctypes front-end of oracle/hashgrid_ref.c (TEST INFRASTRUCTURE; built by `make -C oracle` or
`__graft_entry__.build()`)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libls2fm_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "hashgrid_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def level_table(n_levels, base_resolution, per_level_scale, log2_hashmap_size):
    scale = np.zeros(n_levels, np.float32)
    res = np.zeros(n_levels, np.uint32)
    size = np.zeros(n_levels, np.uint32)
    offset = np.zeros(n_levels + 1, np.uint32)
    hashed = np.zeros(n_levels, np.uint8)
    lib().ls2fm_ref_level_table(n_levels, base_resolution, ctypes.c_float(per_level_scale), log2_hashmap_size,
                                _p(scale, ctypes.c_float), _p(res, ctypes.c_uint32), _p(size, ctypes.c_uint32),
                                _p(offset, ctypes.c_uint32), _p(hashed, ctypes.c_uint8))
    return scale, res, size, offset, hashed.astype(bool)


def grid_indices(x, table, want_weights=False):
    x = np.ascontiguousarray(x, np.float32)
    m, L = x.shape[0], table.n_levels
    idx = np.zeros((m, L, 8), np.uint32)
    w = np.zeros((m, L, 8), np.float32) if want_weights else None
    lib().ls2fm_ref_grid_indices(_p(x, ctypes.c_float), ctypes.c_int64(m), L, _p(table.scale, ctypes.c_float),
                                 _p(table.resolution, ctypes.c_uint32), _p(table.size, ctypes.c_uint32),
                                 _p(idx, ctypes.c_uint32), _p(w, ctypes.c_float) if want_weights else None)
    return (idx, w) if want_weights else idx


def grid_encode(x, params, table, want_dy_dx=False):
    x = np.ascontiguousarray(x, np.float32)
    params = np.ascontiguousarray(params, np.float32)
    m, L, F = x.shape[0], table.n_levels, table.n_features
    out = np.zeros((m, L * F), np.float32)
    dydx = np.zeros((m, L * F, 3), np.float32) if want_dy_dx else None
    lib().ls2fm_ref_grid_encode(_p(x, ctypes.c_float), ctypes.c_int64(m), _p(params, ctypes.c_float), L, F,
                                _p(table.scale, ctypes.c_float), _p(table.resolution, ctypes.c_uint32),
                                _p(table.size, ctypes.c_uint32), _p(table.offset, ctypes.c_uint32),
                                _p(out, ctypes.c_float), _p(dydx, ctypes.c_float) if want_dy_dx else None)
    return (out, dydx) if want_dy_dx else out
