"""probe: how much does L2 residency of a level's table slice matter for the gather pass?  forward-only (no graph), per-kernel
device times from the library profiler, for log2_T in {17, 18, 19} and both storage forms (two tables / interleaved copy)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import randomize, synthetic_rays
from ls2fm import _lib, fused
from ls2fm.profile import kernel_times
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer

lib = _lib.load()
dev = torch.device("cuda", 0)
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for log2_t in (17, 18, 19):
    for mode in ("off", "version"):
        fused._DUAL_TABLE = mode
        opt = make_options("ETH3D", device=str(dev), dual_field=True, sample_intvs=128,
                           hash_encoding=dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=log2_t, base_resolution=16))
        torch.manual_seed(0)
        sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
        randomize([sdf, rad], seed=0)
        center, ray = synthetic_rays(rays, 5.0, dev, seed=0)
        for grad in (False, True):           # grad enabled: the gather pass also counts the scatter's items
            with torch.set_grad_enabled(grad):
                for _ in range(5):
                    ren.forward(opt, center, ray, sdf, rad)
                torch.cuda.synchronize()
                lib.ls2fm_profile_reset(); lib.ls2fm_profile_enable(1)
                for _ in range(30):
                    ren.forward(opt, center, ray, sdf, rad)
                torch.cuda.synchronize()
                lib.ls2fm_profile_enable(0)
            t = kernel_times(lib)
            print(f"T=2^{log2_t} dual_table={mode:8s} counting={grad}", {k: round(v[0], 1) for k, v in t.items() if "encode" in k or "shade" in k}, flush=True)
