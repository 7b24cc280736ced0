"""Radiance field with the reference's class surface (models/RadF.py): optional second hash grid +
Geometry MLP (`dual_field`), Fourier view embedding, radiance decoder."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from ..util_layers import get_layer_dims
from .base import Geometry, Radiance, get_Embedder


class RadF(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        lo = torch.tensor(np.array(opt.data.bound_min), dtype=torch.float32)[None, None, :]
        hi = torch.tensor(np.array(opt.data.bound_max), dtype=torch.float32)[None, None, :]
        self.register_buffer("bound_max", hi, persistent=False)
        self.register_buffer("bound_min", lo, persistent=False)
        self.register_buffer("center", (hi + lo) / 2, persistent=False)
        self.register_buffer("half_size", (hi - lo) / 2, persistent=False)
        self.rescale = opt.SDF.VolSDF.rescale
        self.define_network(opt)

    @property
    def dual_field(self) -> bool:
        return self.opt.Ablate_config.dual_field == True  # noqa: E712

    def define_network(self, opt):
        geo_layers = get_layer_dims(opt.SDF.arch.layers)
        feat_dim = geo_layers[-1][-1]
        if self.dual_field:
            # a second field (own table + Geometry MLP) feeds extra features to the decoder
            self.embed_fn = get_Embedder(opt=opt, input_dim=3, input_choice="Hash")
            self.Geo_enc = Geometry(opt=opt, input_dim=self.embed_fn.out_dim, skip=opt.SDF.arch.skip,
                                    tf_init=opt.SDF.NN_Init.tf_init, layers=geo_layers)
        self.embed_fn_v = get_Embedder(opt=opt, input_dim=3, input_choice="Fourier")
        # decoder input: point(3) + normal(3) + view embedding + geometry feature(s)
        in_dim = 3 + 3 + self.embed_fn_v.out_dim + feat_dim * (2 if self.dual_field else 1)
        self.Rad_dec = Radiance(opt=opt, input_dim=in_dim, skip=opt.SDF.arch.skip, tf_init=opt.SDF.NN_Init.tf_init,
                                layers=get_layer_dims(opt.RadF.arch.layers))

    def Geometry_feat(self, xyz):
        enc = self.embed_fn(xyz, rescale=self.rescale, bound_min=self.bound_min, bound_max=self.bound_max)
        return self.Geo_enc(enc)

    def infer_embed_v(self, ray_utils):
        return self.embed_fn_v(ray_utils)

    def infer_app(self, geo_enc):
        """[..., C] decoder input -> rgb [..., 3]"""
        return self.Rad_dec(geo_enc)
