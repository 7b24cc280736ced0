"""Host-side geometry of the multiresolution hash grid and the autograd-capable encode op.

`build_grid_desc` does what tinycudann's GridEncodingTemplated constructor does for the config the
reference builds in models/base.py:120-139 (grid_scale, grid_resolution, round-up-to-8, cap at
2^log2_hashmap_size, offset table, dense-vs-hash decision of grid_index) and hands the result to the
kernels as an `ls2fm_grid_desc`.  `Encoding` mirrors the tcnn.Encoding module surface the reference
relies on (models/base.py:17, :37; state_dict key `params`).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from .ops import grid_encode


def per_level_scale_from_bounds(bound_min0: float, bound_max0: float, n_levels: int, base_resolution: int) -> float:
    """models/base.py:128-129: b = exp(ln(2048 * s / N_min) / (L - 1)), s = half extent on axis 0."""
    s = (bound_max0 - bound_min0) / 2
    return float(np.exp(np.log(2048 * s / base_resolution) / (n_levels - 1)))


def build_grid_desc(n_levels: int, n_features: int, log2_hashmap_size: int, base_resolution: int,
                    per_level_scale: float) -> _lib.GridDesc:
    if not 1 <= n_levels <= _lib.MAX_LEVELS:
        raise ValueError(f"n_levels must be in [1, {_lib.MAX_LEVELS}] (got {n_levels})")
    if n_features != 2:
        raise ValueError("the kernels implement n_features_per_level = 2 (options/config_hash_sdf.json:5)")
    desc = _lib.GridDesc()
    desc.n_levels, desc.n_features = n_levels, n_features
    b32 = np.float32(per_level_scale)
    # log2 / exp2 evaluated in float64 and rounded once: platform-independent level scales
    log2_b = np.float32(np.log2(np.float64(b32)))
    cap = 1 << log2_hashmap_size
    first = 0
    for level in range(n_levels):
        growth = np.float32(np.exp2(np.float64(np.float32(level) * log2_b)))
        scale = np.float32(growth * np.float32(base_resolution) - np.float32(1.0))
        res = int(math.ceil(float(scale))) + 1
        limit = 0xFFFFFFFF // 2
        entries = limit if float(np.float32(res) ** np.float32(3)) > float(np.float32(limit)) else res ** 3
        entries = min(-(-entries // 8) * 8, cap)
        # tcnn grid_index: walk the dense strides while they fit; hash when the level overflows
        stride, dims = 1, 0
        while dims < 3 and stride <= entries:
            stride *= res
            dims += 1
        desc.scale[level] = float(scale)
        desc.resolution[level] = res
        desc.size[level] = entries
        desc.offset[level] = first
        desc.hashed[level] = 1 if entries < stride else 0
        first += entries
    desc.offset[n_levels] = first
    return desc


def n_table_floats(desc: _lib.GridDesc) -> int:
    return int(desc.offset[desc.n_levels]) * desc.n_features


class Encoding(torch.nn.Module):
    """Drop-in for `tinycudann.Encoding(n_input_dims=3, encoding_config={otype Grid, type Hash,
    interpolation Linear, ...})`: `.n_output_dims`, one flat fp32 Parameter `params` initialised
    U(-1e-4, 1e-4), `forward(x[M,3]) -> [M, L*F]` differentiable to `params` and `x` and twice
    differentiable along the x path.  Output is fp32 (the build's precision; real tcnn returns fp16,
    SURVEY.md C-11)."""

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None):
        super().__init__()
        if n_input_dims != 3:
            raise ValueError("only 3-D positions are used on this path")
        cfg = encoding_config
        if cfg.get("otype", "Grid") not in ("Grid", "HashGrid") or cfg.get("type", "Hash") != "Hash" \
                or cfg.get("interpolation", "Linear") != "Linear":
            raise ValueError(f"unsupported encoding config {cfg}: only Grid/Hash/Linear is on the path")
        self.encoding_config = dict(cfg)
        self.desc = build_grid_desc(int(cfg["n_levels"]), int(cfg["n_features_per_level"]),
                                    int(cfg["log2_hashmap_size"]), int(cfg["base_resolution"]),
                                    float(cfg["per_level_scale"]))
        self.n_input_dims = 3
        self.n_output_dims = self.desc.n_levels * self.desc.n_features
        gen = torch.Generator().manual_seed(seed)
        init = (torch.rand(n_table_floats(self.desc), generator=gen) * 2 - 1) * 1e-4
        self.params = torch.nn.Parameter(init.float())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return grid_encode(x, self.params, self.desc)

    def extra_repr(self) -> str:
        d = self.desc
        return f"levels={d.n_levels}, features=2, params={n_table_floats(d)}, top_res={d.resolution[d.n_levels - 1]}"
