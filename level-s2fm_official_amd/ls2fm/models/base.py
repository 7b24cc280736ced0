"""Building blocks with the reference's names and state_dict layout (models/base.py):

Embedder_Hash (:12-40)      hash-grid encode of the normalised position, raw xyz prepended
Embedder_Fourier (:43-97)   4-octave sin/cos view embedding (27 dims)
get_Embedder (:101-160)     builds either, deriving per_level_scale from the scene bounds
Geometry (:164-217)         weight-normed MLP, geometric init, Softplus(beta=100)
Radiance (:221-261)         weight-normed MLP ending in a sigmoid (no hidden activation: SURVEY C-1)

These modules are the *general* (autograd-composed) form of the path: the hash grid runs in the HIP
kernels through ls2fm.ops, the tiny dense layers through torch.  The fused kernels used by
Renderer.forward / SDF.infer_sdf read the same Parameters directly.
"""
from __future__ import annotations

import json
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hashgrid
from ..util_layers import get_layer_dims  # noqa: F401  (re-exported for callers that expect it here)

try:                                      # legacy parametrisation keeps the weight_g / weight_v keys
    from torch.nn.utils import weight_norm as _weight_norm
except ImportError:                       # pragma: no cover
    _weight_norm = None


class Embedder_Hash(nn.Module):
    def __init__(self, kwargs, include_input=True, input_dim=3):
        super().__init__()
        self.embedder_obj = hashgrid.Encoding(n_input_dims=input_dim, encoding_config=kwargs)
        self.input_dim = input_dim
        self.include_input = include_input
        self.out_dim = self.embedder_obj.n_output_dims + input_dim

    def forward(self, input, bound_min, bound_max, rescale: float = 1.0):
        if input.shape[-1] != self.input_dim:
            raise ValueError(f"expected [..., {self.input_dim}] positions, got {tuple(input.shape)}")
        lo = bound_min.to(input.device)
        unit = (input - lo) / (bound_max.to(input.device) - lo)           # -> [0,1] inside the scene box
        enc = self.embedder_obj(unit.reshape(-1, self.input_dim)).view(*input.shape[:-1], -1)
        if not self.include_input:
            return enc
        return torch.cat([input / rescale, enc], dim=-1)


class Embedder_Fourier(nn.Module):
    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True,
                 periodic_fns=(torch.sin, torch.cos)):
        super().__init__()
        self.input_dim = input_dim
        self.include_input = include_input
        self.periodic_fns = tuple(periodic_fns)
        if log_sampling:
            bands = [2.0 ** (max_freq_log2 * i / (N_freqs - 1)) if N_freqs > 1 else 1.0 for i in range(N_freqs)]
        else:
            hi = 2.0 ** max_freq_log2
            bands = [1.0 + (hi - 1.0) * i / (N_freqs - 1) if N_freqs > 1 else 1.0 for i in range(N_freqs)]
        self.freq_bands = [float(b) for b in bands]
        self.out_dim = input_dim * (int(include_input) + N_freqs * len(self.periodic_fns))

    def forward(self, input, bound_min=None, bound_max=None, rescale: float = 1.0):
        if input.shape[-1] != self.input_dim:
            raise ValueError(f"expected [..., {self.input_dim}] directions, got {tuple(input.shape)}")
        parts = [input / rescale] if self.include_input else []
        for band in self.freq_bands:
            parts.extend(fn(input * band) for fn in self.periodic_fns)
        return torch.cat(parts, dim=-1)


def _hash_encoding_config(opt):
    hc = opt.SDF.Hash_config
    inline = hc.get("encoding", None) if isinstance(hc, dict) else getattr(hc, "encoding", None)
    if inline is not None:
        return dict(inline)
    with open(hc.config_file) as f:
        return json.load(f)["encoding"]


def get_Embedder(opt, input_dim=3, input_choice="Hash", choices=("Hash", "Fourier", "SH")):
    if input_choice not in choices:
        raise ValueError(f"Invalid input option. Valid choices are: {choices}")
    if input_choice == "Hash":
        enc = _hash_encoding_config(opt)
        levels, base = int(enc["n_levels"]), int(enc["base_resolution"])
        # the JSON's own per_level_scale is ignored: it is derived from the scene extent (base.py:128-129)
        growth = hashgrid.per_level_scale_from_bounds(opt.data.bound_min[0], opt.data.bound_max[0], levels, base)
        return Embedder_Hash(kwargs={
            "otype": "Grid", "type": "Hash", "n_levels": levels,
            "n_features_per_level": int(enc["n_features_per_level"]),
            "log2_hashmap_size": int(enc["log2_hashmap_size"]), "base_resolution": base,
            "per_level_scale": growth, "interpolation": "Linear"}, input_dim=input_dim)
    if input_choice == "Fourier":
        return Embedder_Fourier(input_dim=input_dim, max_freq_log2=3, N_freqs=4, log_sampling=True,
                                include_input=True, periodic_fns=(torch.sin, torch.cos))
    raise NotImplementedError("the SphericalHarmonics branch is never selected by the reference (SURVEY.md section 0)")


def _normed(linear: nn.Linear) -> nn.Module:
    if _weight_norm is None:
        raise RuntimeError("torch.nn.utils.weight_norm is required for state_dict compatibility")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _weight_norm(linear)


class Geometry(nn.Module):
    """in -> hidden -> ... -> (feat + 1) with Softplus(100) between layers; channel 0 of the output is the
    (unsigned, unscaled) SDF.  Geometric initialisation (sphere of radius `bias`) when tf_init."""

    def __init__(self, opt, input_dim, layers, skip=(), tf_init=True):
        super().__init__()
        self.skip = list(skip)
        self.mlp = nn.ModuleList()
        last = len(layers) - 1
        sphere_radius = opt.SDF.NN_Init.bias
        for li, (k_in, k_out) in enumerate(layers):
            k_in = input_dim if li == 0 else k_in
            if li in self.skip:
                k_in += input_dim
            if li == last:
                k_out += 1
            lin = nn.Linear(k_in, k_out)
            if tf_init:
                self._geometric_init(lin, li, last, layers, input_dim, sphere_radius)
            self.mlp.append(_normed(lin))
        self.softplus = nn.Softplus(beta=100, threshold=20)

    def _geometric_init(self, lin, li, last, layers, input_dim, radius):
        with torch.no_grad():
            k_out = lin.out_features
            if li == last:
                lin.weight.normal_(mean=math.sqrt(math.pi) / math.sqrt(layers[li][0]), std=1e-4)
                lin.bias.fill_(-radius)
                return
            lin.bias.zero_()
            lin.weight.normal_(0.0, math.sqrt(2) / math.sqrt(k_out))
            if li == 0:
                lin.weight[:, 3:] = 0.0             # only xyz drives the initial sphere
            elif li in self.skip:
                lin.weight[:, -(input_dim - 3):] = 0.0

    def forward(self, points_enc):
        feat = points_enc
        n = len(self.mlp)
        for li, layer in enumerate(self.mlp):
            if li in self.skip:
                feat = torch.cat([feat, points_enc], dim=-1) / math.sqrt(2)
            feat = layer(feat)
            if li < n - 1:
                feat = self.softplus(feat)
        return feat


class Radiance(nn.Module):
    """Radiance decoder.  The reference's hidden ReLU never fires (its test reads the length of an empty
    ModuleList: models/base.py:255-258, SURVEY C-1), so the decoder is affine-affine-affine-sigmoid; that
    behaviour is reproduced (set `hidden_relu=True` to get the presumably intended network instead)."""

    def __init__(self, opt, input_dim, layers, skip=(), tf_init=True, hidden_relu=False):
        super().__init__()
        self.opt = opt
        self.skip = list(skip)
        self.hidden_relu = hidden_relu
        self.mlp = nn.ModuleList()            # kept (empty) for state_dict / attribute parity
        self.mlp_radiance = nn.ModuleList()
        for li, (k_in, k_out) in enumerate(layers):
            lin = nn.Linear(input_dim if li == 0 else k_in, k_out)
            self.mlp_radiance.append(_normed(lin) if tf_init else lin)

    def forward(self, geo_enc):
        feat = geo_enc
        n = len(self.mlp_radiance)
        for li, layer in enumerate(self.mlp_radiance):
            feat = layer(feat)
            if self.hidden_relu and li < n - 1:
                feat = F.relu(feat)
        return torch.sigmoid(feat)
