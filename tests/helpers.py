"""Shared helpers for the product-side tests (build the ls2fm classes for a golden case, etc.)."""
import numpy as np
import torch

from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer

DS_ALIAS = {"DTU": "DTU", "ETH3D": "ETH3D", "BlendedMVS": "BlendedMVS", "scannet": "scannet"}


def options_for(meta, device):
    opt = make_options(DS_ALIAS[meta["dataset"]], device=device, dual_field=meta["dual_field"],
                       sample_intvs=meta["n_samples"],
                       hash_encoding=dict(n_levels=meta["n_levels"], n_features_per_level=2,
                                          log2_hashmap_size=meta["log2_hashmap_size"], base_resolution=16))
    opt.data.bg_sdf = meta["bg_sdf"]
    opt.data.bgcolor = list(meta["bgcolor"])
    opt.SDF.VolSDF.iters_max_st = meta["iters_max_st"]
    return opt


def product_for(meta, golden, device, sdf_prefix="sdf"):
    """ls2fm SDF / RadF / Renderer for a golden case with the reference-captured weights loaded"""
    opt = options_for(meta, device)
    sdf, rad, ren = SDF(opt).to(device), RadF(opt).to(device), Renderer(opt)
    sd = {k[len(sdf_prefix) + 1:]: torch.from_numpy(v) for k, v in golden.items() if k.startswith(sdf_prefix + "/")}
    rd = {k[4:]: torch.from_numpy(v) for k, v in golden.items() if k.startswith("rad/")}
    missing, unexpected = sdf.load_state_dict(sd, strict=True), None
    rad.load_state_dict(rd, strict=True)
    return opt, sdf, rad, ren


def named_grads(module):
    return {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu())
            for k, p in module.named_parameters()}


def oracle_table_of(desc):
    """oracle LevelTable equivalent of a product GridDesc (for feeding the oracle the same geometry)"""
    from oracle.hashgrid import LevelTable
    L = desc.n_levels
    return LevelTable(L, 2, 16, 0.0, 0,
                      np.array(desc.scale[:L], np.float32), np.array(desc.resolution[:L], np.uint32),
                      np.array(desc.size[:L], np.uint32), np.array(desc.offset[:L + 1], np.uint32),
                      np.array(desc.hashed[:L], bool))
