"""Oracle: multiresolution hash-grid encoding (tcnn ``Grid`` / ``Hash`` / ``Linear``).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference:
tiny-cuda-nn 1.7 is a third-party dependency whose source is not in
/root/reference (call sites: models/base.py:17 constructor, models/base.py:37
forward).  This file restates the published tcnn algorithm:

  * level geometry  - tcnn ``grid_scale`` / ``grid_resolution`` and the
    ``GridEncodingTemplated`` constructor (offset table, round-up-to-8, min with
    2^log2_hashmap_size); SURVEY.md Appendix A.2
  * lookup          - tcnn ``kernel_grid``: ``pos = fmaf(scale, x, 0.5)``, floor,
    uint32 wrap, 8-corner ``grid_index`` (dense stride walk, coherent prime hash
    {1, 2654435761, 805459861} when the level does not fit, ``% hashmap_size``),
    tri-linear blend, level-major output columns ``[l*F + f]``
  * the config the reference builds for it   - models/base.py:120-139

All floating point runs in the dtype of ``x`` (float32 for parity runs, float64
for derivative checks); index arithmetic is exact integer arithmetic.  The
function is built from differentiable torch ops so first and second derivatives
w.r.t. ``x`` and ``params`` come from autograd (the reference needs
``create_graph=True`` through this op, models/SDF.py:102-114).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861
_U32 = 0xFFFFFFFF


@dataclass
class LevelTable:
    """Per-level geometry of one hash grid (all host-side, numpy)."""

    n_levels: int
    n_features: int
    base_resolution: int
    per_level_scale: float      # as float32 value
    log2_hashmap_size: int
    scale: np.ndarray           # float32 [L]
    resolution: np.ndarray      # uint32  [L]
    size: np.ndarray            # uint32  [L]  entries in the level
    offset: np.ndarray          # uint32  [L+1] prefix sum of size (in entries)
    hashed: np.ndarray          # bool    [L]  level uses the prime hash

    @property
    def n_params(self) -> int:
        return int(self.offset[-1]) * self.n_features

    @property
    def n_output_dims(self) -> int:
        return self.n_levels * self.n_features


def reference_per_level_scale(bound_min0: float, bound_max0: float, n_levels: int, base_resolution: int) -> float:
    """per_level_scale exactly as the reference derives it (models/base.py:128-129):
    ``b = exp(ln(2048*s/N_min)/(L-1))`` with ``s`` the scene half extent on axis 0
    (double precision on the host; tcnn then stores it as a float)."""
    s = (bound_max0 - bound_min0) / 2
    return float(np.exp(np.log(2048 * s / base_resolution) / (n_levels - 1)))


def make_level_table(n_levels: int, n_features: int, log2_hashmap_size: int, base_resolution: int,
                     per_level_scale: float, n_pos_dims: int = 3) -> LevelTable:
    """tcnn level geometry in float32 (grid_scale / grid_resolution / offset table)."""
    # tcnn evaluates log2f / exp2f on the device (CUDA: <= 2 ulp); vendor libms disagree in the last
    # bit (numpy vs glibc differ at some levels), so the oracle DEFINES the scale with correctly
    # rounded log2/exp2 (evaluated in float64, rounded once to float32).  All other steps are the
    # float32 operations of grid_scale().  The product passes this host-built table to the device.
    b = np.float32(per_level_scale)
    log2b = np.float32(np.log2(np.float64(b)))
    scale = np.zeros(n_levels, np.float32)
    res = np.zeros(n_levels, np.uint32)
    size = np.zeros(n_levels, np.uint32)
    offset = np.zeros(n_levels + 1, np.uint32)
    hashed = np.zeros(n_levels, bool)
    max_params = np.uint32(0xFFFFFFFF // 2)
    off = 0
    for l in range(n_levels):
        e = np.float32(np.exp2(np.float64(np.float32(l) * log2b)))
        s = np.float32(e * np.float32(base_resolution) - np.float32(1.0))
        r = int(np.ceil(s)) + 1
        if np.float32(r) ** np.float32(n_pos_dims) > np.float32(max_params):
            n = int(max_params)
        else:
            n = r ** n_pos_dims
        n = (n + 7) // 8 * 8
        dense_n = n
        n = min(n, 1 << log2_hashmap_size)
        scale[l] = s
        res[l] = r
        size[l] = n
        # the level is hashed iff the dense stride walk overflows the level (grid_index)
        stride = 1
        for _ in range(n_pos_dims):
            if stride > n:
                break
            stride *= r
        hashed[l] = n < stride
        offset[l] = off
        off += n
        del dense_n
    offset[n_levels] = off
    return LevelTable(n_levels, n_features, base_resolution, float(b), log2_hashmap_size,
                      scale, res, size, offset, hashed)


def _pos_floor(x: torch.Tensor, scale: float):
    """``pos = fmaf(scale, x, 0.5)`` (single rounding), its floor as uint32-wrapped
    int64, and the fractional part carrying d/dx = scale for autograd."""
    if x.dtype == torch.float32:
        # a float32*float32 product is exact in float64; adding 0.5 and rounding once to
        # float32 reproduces fmaf up to a 2^-29-probability double rounding (the C oracle,
        # which calls fmaf, is the index authority: tests compare the two).
        pos = (x.detach().double() * float(scale) + 0.5).float()
    else:
        pos = x.detach() * float(scale) + 0.5
    fl = torch.floor(pos)
    cell = fl.to(torch.int64) & _U32            # (uint32)(int)floorf(pos)
    frac_const = pos - fl
    # value == frac_const exactly (x - x.detach() is an exact zero); derivative == scale
    frac = frac_const + float(scale) * (x - x.detach())
    return cell, frac


def corner_indices(cell: torch.Tensor, res: int, size: int, hashed: bool) -> torch.Tensor:
    """Entry index (within the level) of the 8 cell corners -> int64 [M, 8].

    Corner id bit d selects ``cell[d] + 1`` on axis d (tcnn kernel_grid loop order)."""
    out = []
    for corner in range(8):
        c = [(cell[:, d] + ((corner >> d) & 1)) & _U32 for d in range(3)]
        if hashed:
            idx = (c[0] ^ ((c[1] * PRIME_Y) & _U32) ^ ((c[2] * PRIME_Z) & _U32)) & _U32
        else:
            stride = 1
            idx = torch.zeros_like(c[0])
            for d in range(3):
                if stride > size:
                    break
                idx = (idx + c[d] * stride) & _U32
                stride = (stride * res) & _U32
        out.append(idx % size)
    return torch.stack(out, dim=1)


def grid_indices(x: torch.Tensor, table: LevelTable) -> torch.Tensor:
    """All corner entry indices (level-local) -> int64 [M, L, 8]."""
    per_level = []
    for l in range(table.n_levels):
        cell, _ = _pos_floor(x, float(table.scale[l]))
        per_level.append(corner_indices(cell, int(table.resolution[l]), int(table.size[l]), bool(table.hashed[l])))
    return torch.stack(per_level, dim=1)


def encode(x: torch.Tensor, params: torch.Tensor, table: LevelTable) -> torch.Tensor:
    """Hash-grid encode.  x [M,3] (the reference feeds (p-bmin)/(bmax-bmin); no clamping),
    params flat [n_params] -> [M, L*F]."""
    assert x.dim() == 2 and x.shape[1] == 3
    F = table.n_features
    grid = params.view(-1, F)
    cols = []
    for l in range(table.n_levels):
        cell, w = _pos_floor(x, float(table.scale[l]))
        idx = corner_indices(cell, int(table.resolution[l]), int(table.size[l]), bool(table.hashed[l]))
        idx = idx + int(table.offset[l])
        acc = torch.zeros(x.shape[0], F, dtype=x.dtype, device=x.device)
        for corner in range(8):
            wt = torch.ones_like(w[:, 0])
            for d in range(3):
                wt = wt * (w[:, d] if (corner >> d) & 1 else (1 - w[:, d]))
            acc = acc + wt[:, None] * grid[idx[:, corner]].to(x.dtype)
        cols.append(acc)
    return torch.cat(cols, dim=1)


class OracleEncoding(torch.nn.Module):
    """nn.Module with tcnn.Encoding's observable surface (``n_output_dims``, one flat
    fp32 Parameter ``params`` initialised U(-1e-4, 1e-4), ``forward(x[M,3]) -> [M, L*F]``).
    Used (a) by the oracle's own fields and (b) as the stand-in handed to the imported
    reference when golden vectors are generated."""

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None):
        super().__init__()
        assert n_input_dims == 3
        assert encoding_config.get("otype", "Grid") in ("Grid", "HashGrid")
        assert encoding_config.get("type", "Hash") == "Hash"
        assert encoding_config.get("interpolation", "Linear") == "Linear"
        self.table = make_level_table(
            n_levels=int(encoding_config["n_levels"]),
            n_features=int(encoding_config["n_features_per_level"]),
            log2_hashmap_size=int(encoding_config["log2_hashmap_size"]),
            base_resolution=int(encoding_config["base_resolution"]),
            per_level_scale=float(encoding_config["per_level_scale"]),
        )
        self.n_input_dims = 3
        self.n_output_dims = self.table.n_output_dims
        g = torch.Generator().manual_seed(seed)
        init = (torch.rand(self.table.n_params, generator=g) * 2 - 1) * 1e-4
        self.params = torch.nn.Parameter(init.float())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return encode(x, self.params, self.table)
