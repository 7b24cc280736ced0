"""The slice of the reference's `opt` object that the hot path reads, as an attribute dict.

The reference passes one EasyDict built by utils/options.py from options/*.yaml everywhere; its config
plumbing is out of scope here, but the classes of this package must accept
the very same object.  They only touch the keys listed in SURVEY.md section 5 ("config / flags"), so
any attribute-style mapping works: the reference's EasyDict, or the `Options` built below, which
re-states those keys with the values of options/LevelS2fM.yaml + the per-dataset yaml.
"""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_HASH_CONFIG = os.path.join(_HERE, "configs", "config_hash_sdf.json")


class Options(dict):
    """dict with attribute access, recursive (EasyDict-compatible for the keys used on the path)"""

    def __init__(self, mapping=None, **kw):
        super().__init__()
        for k, v in dict(mapping or {}, **kw).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, Options):
            value = Options(value)
        super().__setitem__(key, value)

    def __setattr__(self, key, value):
        self[key] = value

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None


# per-dataset switches that change the arithmetic (second table)
DATASET_PRESETS = {
    # name: bounds half extent, inside, scale_mlp, init bias, iters_max_st, bgcolor
    "DTU": dict(s=1.0, inside=True, scale_mlp=1.0, bias=0.5, iters_max_st=10, bgcolor=[0, 0, 0]),
    "ETH3D": dict(s=5.0, inside=False, scale_mlp=5.0, bias=2.5, iters_max_st=20, bgcolor=[0, 0, 0]),
    "BlendedMVS": dict(s=2.0, inside=True, scale_mlp=3.0, bias=1.0, iters_max_st=20, bgcolor=[1, 1, 1]),
    "scannet": dict(s=4.0, inside=False, scale_mlp=1.0, bias=2.0, iters_max_st=10, bgcolor=[0, 0, 0]),
}


def make_options(dataset: str = "DTU", device: str = "cuda", dual_field: bool = False, sample_intvs: int = 128,
                 rand_rays: int = 8192, scene: str = "scene0", hash_encoding: dict | None = None, **data_overrides):
    """Options for one of the reference's four dataset presets.  `hash_encoding` optionally replaces the
    JSON file (keys n_levels, n_features_per_level, log2_hashmap_size, base_resolution)."""
    p = DATASET_PRESETS[dataset]
    s = p["s"]
    opt = Options(
        device=device, Res=100,
        Ablate_config=dict(dual_field=dual_field),
        SDF=dict(
            arch=dict(layers=[None, 64, 16], skip=[]),
            NN_Init=dict(scale_mlp=p["scale_mlp"], bias=p["bias"], tf_init=True),
            VolSDF=dict(sample_intvs=sample_intvs, final_sample_intvs=64, volsdf_sampling=False,
                        iters_max_st=p["iters_max_st"], eps=0.1, beta_init=0.05, rescale=1.0, beta_speed=1.0,
                        sdf_threshold=1e-3, max_upsample_iter=6),
            Hash_config=dict(config_file=DEFAULT_HASH_CONFIG),
        ),
        RadF=dict(arch=dict(layers=[None, 64, 64, 3], skip=[])),
        Renderer=dict(rand_rays=rand_rays),
        data=dict(dataset=dataset, scene=scene, inside=p["inside"], bg_sdf=None, bg_rad=2, bgcolor=list(p["bgcolor"]),
                  bound_min=[-s, -s, -s], bound_max=[s, s, s]),
    )
    if hash_encoding is not None:
        opt.SDF.Hash_config["encoding"] = dict(hash_encoding)
    opt.data[scene] = Options()
    for k, v in data_overrides.items():
        opt.data[k] = v
    return opt
