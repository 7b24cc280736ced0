#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE's own classes.

Runs only in the build container (needs /root/reference; never on the GPU box).  Usage:

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, manifest.json

What is imported from /root/reference: models/{SDF,RadF,Renderer,base}.py, utils/{camera,
custom_functions,util}.py -- unmodified, from where they lie.  Third-party modules the image
lacks are pre-seeded in sys.modules:

  * pure import shims, never executed on the path: easydict (attr-dict), ipdb, termcolor,
    plyfile, skimage, open3d, torch_scatter
  * the two third-party CUDA ops whose sources are not in the reference tree are supplied by
    the oracle (so they are NOT pinned by these vectors -- see oracle/__init__.py):
        tinycudann.Encoding        -> oracle.hashgrid.OracleEncoding
        vren.ray_aabb_intersect    -> oracle.ray_aabb.ray_aabb_intersect (returns a list)
  * torch.Tensor.cuda is made the identity because SDF.sphere_tracing hard-codes .cuda()
    (models/SDF.py:125-155).

Everything recorded here is data (inputs, weights, outputs, gradients); no reference source
text is written anywhere.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import hashgrid as o_hash          # noqa: E402
from oracle import ray_aabb as o_aabb          # noqa: E402
import losses                                   # noqa: E402


class AttrDict(dict):
    """minimal stand-in for easydict.EasyDict"""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            self[k] = v


def install_stubs():
    def shim(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    shim("easydict", EasyDict=AttrDict)
    shim("ipdb", set_trace=lambda *a, **k: None)
    shim("termcolor", colored=lambda s, *a, **k: s)
    shim("plyfile")
    shim("skimage")
    shim("open3d")
    shim("torch_scatter", segment_csr=None)
    shim("vren", ray_aabb_intersect=o_aabb.ray_aabb_intersect)
    shim("tinycudann", Encoding=o_hash.OracleEncoding)
    torch.Tensor.cuda = lambda self, *a, **k: self


def make_opt(dataset, hash_json, dual_field, n_samples, bg_sdf=None, bgcolor=None, iters_max=None):
    from oracle.fields import DATASETS
    ds = DATASETS[dataset]
    scene = "scene0"
    opt = AttrDict(
        device="cpu", Res=100, H=4, W=4,
        Ablate_config=dict(dual_field=dual_field),
        SDF=dict(
            arch=dict(layers=[None, 64, 16], skip=[]),
            NN_Init=dict(scale_mlp=ds["scale_mlp"], bias=ds["bias"], tf_init=True),
            VolSDF=dict(sample_intvs=n_samples, volsdf_sampling=False,
                        iters_max_st=iters_max if iters_max is not None else ds["iters_max_st"],
                        beta_init=0.05, rescale=1.0, beta_speed=1.0, sdf_threshold=1e-3),
            Hash_config=dict(config_file=hash_json),
        ),
        RadF=dict(arch=dict(layers=[None, 64, 64, 3], skip=[])),
        data=dict(dataset=dataset, scene=scene, inside=ds["inside"], bg_sdf=bg_sdf, bg_rad=2,
                  bgcolor=list(bgcolor if bgcolor is not None else ds["bgcolor"]),
                  bound_min=list(ds["bound_min"]), bound_max=list(ds["bound_max"])),
    )
    opt.data[scene] = AttrDict()
    return opt


def write_hash_json(n_levels, log2_T, base=16):
    fd, path = tempfile.mkstemp(suffix=".json")
    with os.fdopen(fd, "w") as f:
        json.dump({"encoding": {"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": 2,
                                "log2_hashmap_size": log2_T, "base_resolution": base,
                                "per_level_scale": 1.38}}, f)
    return path


def randomize_module(mod, gen, table_amp=0.1, w_std=0.05):
    """non-degenerate weights (SURVEY C-12): tables U(-amp,amp); hash columns of every Geometry
    first layer N(0,w_std); weight_g refreshed to the row norms."""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if name.endswith("embedder_obj.params"):
                p.copy_((torch.rand(p.shape, generator=gen) * 2 - 1) * table_amp)
            if name.endswith("mlp.0.weight_v"):
                p[:, 3:] = torch.randn(p[:, 3:].shape, generator=gen) * w_std
        for name, p in mod.named_parameters():
            if name.endswith("weight_g"):
                v = dict(mod.named_parameters())[name.replace("weight_g", "weight_v")]
                # perturb g away from ||v|| so the weight-norm scale is exercised
                p.copy_(v.norm(dim=1, keepdim=True) * (1 + 0.1 * torch.randn(p.shape, generator=gen)))
        if hasattr(mod, "beta"):
            mod.beta.add_(0.3)


def sd_np(mod, prefix):
    return {f"{prefix}/{k}": v.detach().numpy().copy() for k, v in mod.state_dict().items()}


def grads_np(mod, prefix):
    out = {}
    for k, p in mod.named_parameters():
        out[f"{prefix}/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    return out


def make_rays(gen, n, s, miss=4, inside=4):
    """rays towards the scene box from z = -2.5 s (unnormalised, as utils/camera.py:246-251 yields),
    plus a few that miss the box and a few whose origin is inside it (near clamps to 0)."""
    c = torch.tensor([0.0, 0.0, -2.5 * s]).repeat(n, 1) + 0.05 * s * torch.randn(n, 3, generator=gen)
    d = torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1) + 0.15 * torch.randn(n, 3, generator=gen)
    d[:miss] = torch.tensor([0.0, 1.0, -0.2]) + 0.05 * torch.randn(miss, 3, generator=gen)     # misses
    c[miss:miss + inside] = 0.3 * s * torch.randn(inside, 3, generator=gen)                      # inside the box
    return c.float(), d.float()


def main():
    assert os.path.isdir(REF), "golden vectors can only be generated where /root/reference exists"
    install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from models.SDF import SDF
    from models.RadF import RadF
    from models.Renderer import Renderer
    from models import base as ref_base

    manifest = {}

    # ------------------------------------------------------------------ R1: Fourier view embedding
    gen = torch.Generator().manual_seed(101)
    d = torch.randn(33, 3, generator=gen)
    emb = ref_base.get_Embedder(opt=None, input_dim=3, input_choice="Fourier")
    np.savez_compressed(os.path.join(HERE, "fourier.npz"), d=d.numpy(), out=emb(d).numpy())

    # ------------------------------------------------------------------ per-dataset field cases
    cases = [
        # name, dataset, L, log2_T, dual, N, bg_sdf, bgcolor
        ("dtu_single", "DTU", 8, 10, False, 16, None, None),
        ("eth3d_dual", "ETH3D", 6, 10, True, 16, None, None),
        ("bmvs_dual_white", "BlendedMVS", 4, 11, True, 12, None, (1, 1, 1)),
        ("dtu_bgsdf", "DTU", 4, 10, False, 8, True, (1, 1, 1)),
        ("scannet_single", "scannet", 5, 10, False, 20, None, None),
    ]
    for ci, (name, dataset, L, log2_T, dual, N, bg_sdf, bgcolor) in enumerate(cases):
        torch.manual_seed(1000 + ci)
        gen = torch.Generator().manual_seed(2000 + ci)
        hash_json = write_hash_json(L, log2_T)
        opt = make_opt(dataset, hash_json, dual, N, bg_sdf=bg_sdf, bgcolor=bgcolor)
        sdf = SDF(opt)
        rad = RadF(opt)
        ren = Renderer(opt)
        s = (opt.data.bound_max[0] - opt.data.bound_min[0]) / 2
        out = {}
        # ---- R8a: sphere tracing at the geometric initialisation (an exact-ish sphere): the loop
        #      ends through the 'every start ray finished' branch with trips < iters_max
        out.update(sd_np(sdf, "sdf_init"))
        c0, d0 = make_rays(gen, 32, s)
        torch.manual_seed(5)
        iters_cfg = sdf.iters_max
        sdf.iters_max = 200
        d_pred0, sdf_last0, sampled0, fmask0 = sdf.sphere_tracing(c0.view(1, -1, 3), d0.view(1, -1, 3), sdf)
        losses.tracing_loss(d_pred0, sdf_last0).backward()
        K0 = (sampled0.shape[1] - 32) // 32
        sdf.iters_max = iters_cfg
        out["st0_iters_max"] = np.int32(200)
        out.update({"st0_center": c0.numpy(), "st0_ray": d0.numpy(), "st0_d_pred": d_pred0.detach().numpy(),
                    "st0_sdf_last": sdf_last0.detach().numpy(), "st0_finish": fmask0.numpy(),
                    "st0_trips": np.int32(K0)})
        out.update(grads_np(sdf, "st0_grad/sdf"))
        sdf.zero_grad()
        randomize_module(sdf, gen)
        randomize_module(rad, gen)
        out.update(sd_np(sdf, "sdf"))
        out.update(sd_np(rad, "rad"))
        meta = dict(dataset=dataset, n_levels=L, log2_hashmap_size=log2_T, dual_field=dual, n_samples=N,
                    bg_sdf=bg_sdf, bgcolor=list(opt.data.bgcolor), iters_max_st=opt.SDF.VolSDF.iters_max_st)

        # ---- R3/R4: infer_sdf modes, gradient, and a scalar of the gradient back-propagated
        pts = (torch.rand(40, 3, generator=gen) * 2 - 1) * s * 1.05     # a few points outside the box
        sdf.zero_grad()
        y, feat = sdf.infer_sdf(pts.clone(), mode="ret_all")
        p_req = pts.clone()
        nrm = sdf.gradient(p_req)
        eik = ((nrm.norm(dim=-1) - 1) ** 2).mean() + 0.2 * (nrm * torch.tensor([0.3, -0.5, 0.8])).sum(-1).mean()
        (eik + 0.1 * y.mean() + 0.05 * (feat[..., 1:] ** 2).mean()).backward()
        out.update({"pts": pts.numpy(), "pts_sdf": y.detach().numpy(), "pts_feat": feat.detach().numpy(),
                    "pts_normal": nrm.detach().numpy(), "pts_dx": p_req.grad.numpy().copy()})
        out.update(grads_np(sdf, "pts_grad/sdf"))

        # ---- R9: get_surface_pts
        sdf.zero_grad()
        p2 = pts.clone()
        surf, nlen = sdf.get_surface_pts(p2)
        out.update({"surf_pts": surf.detach().numpy(), "surf_nlen": nlen.detach().numpy()})

        # ---- R5: forward_ab / sdf_to_sigma (+ d beta)
        sdf.zero_grad()
        a_, b_ = sdf.forward_ab()
        sv = torch.linspace(-0.3, 0.3, 25)[:, None]
        sig = sdf.sdf_to_sigma(sv, a_, b_)
        sig.sum().backward()
        out.update({"ab": np.array([a_.item(), b_.item()], np.float32), "sigma_in": sv.numpy(),
                    "sigma_out": sig.detach().numpy(), "sigma_dbeta": sdf.beta.grad.numpy().copy()})

        # ---- R2 (dual): RadF.Geometry_feat
        if dual:
            out["pts_geofeat"] = rad.Geometry_feat(pts.clone()).detach().numpy()

        # ---- R7: full Renderer.forward + gradients of one scalar to every parameter and to center/ray
        R = 24
        c, dd = make_rays(gen, 2 * R, s)
        center = c.view(2, R, 3).clone().requires_grad_(True)
        ray = dd.view(2, R, 3).clone().requires_grad_(True)
        rgb_t = torch.rand(2, R, 3, generator=gen)
        nm_dir = torch.randn(3, generator=gen)
        sdf.zero_grad(); rad.zero_grad()
        ret = Renderer.forward(ren, opt=opt, center=center, ray=ray, SDF_Field=sdf, Rad_Field=rad)
        loss = losses.render_loss(ret, rgb_t, nm_dir)
        loss.backward()
        out.update({"center": center.detach().numpy(), "ray": ray.detach().numpy(), "rgb_target": rgb_t.numpy(),
                    "nm_dir": nm_dir.numpy(), "render_loss": np.float32(loss.item()),
                    "d_center": center.grad.numpy().copy(), "d_ray": ray.grad.numpy().copy()})
        for k, v in ret.items():
            out[f"ret/{k}"] = v.detach().numpy()
        out.update(grads_np(sdf, "render_grad/sdf"))
        out.update(grads_np(rad, "render_grad/rad"))

        # ---- R6: composite on its own, incl. zero-length rays / misses (t = -1 everywhere)
        gen2 = torch.Generator().manual_seed(3000 + ci)
        rgb_s = torch.rand(1, 6, N, 3, generator=gen2, requires_grad=True)
        sig_s = (torch.rand(1, 6, N, generator=gen2) * 30).requires_grad_(True)
        t_s = torch.sort(torch.rand(1, 6, N, 1, generator=gen2) * 3, dim=2).values
        t_s[0, 0] = -1.0
        rr = torch.randn(1, 6, 3, generator=gen2)
        rr[0, 1] = 0.0
        crgb, cprob = ren.composite(ray=rr, rgb_samples=rgb_s, density_samples=sig_s, depth_samples=t_s)
        (crgb.sum() + (cprob[..., 0] * torch.arange(N - 1)).sum()).backward()
        out.update({"comp_ray": rr.numpy(), "comp_rgb_s": rgb_s.detach().numpy(), "comp_sig_s": sig_s.detach().numpy(),
                    "comp_t_s": t_s.numpy(), "comp_rgb": crgb.detach().numpy(), "comp_prob": cprob.detach().numpy(),
                    "comp_d_rgb_s": rgb_s.grad.numpy().copy(), "comp_d_sig_s": sig_s.grad.numpy().copy()})

        # ---- R8: sphere tracing (RNG-dependent sampled_pts not recorded; its shape is)
        sdf.zero_grad()
        c2, d2 = make_rays(gen, 48, s)
        torch.manual_seed(7)
        d_pred, sdf_last, sampled, fmask = sdf.sphere_tracing(c2.view(1, -1, 3), d2.view(1, -1, 3), sdf)
        losses.tracing_loss(d_pred, sdf_last).backward()
        K = (sampled.shape[1] - 48) // 48
        out.update({"st_center": c2.numpy(), "st_ray": d2.numpy(), "st_d_pred": d_pred.detach().numpy(),
                    "st_sdf_last": sdf_last.detach().numpy(), "st_finish": fmask.numpy(),
                    "st_trips": np.int32(K), "st_sampled_shape": np.array(sampled.shape, np.int32)})
        out.update(grads_np(sdf, "st_grad/sdf"))

        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        manifest[name] = meta
        os.unlink(hash_json)
        print(f"[golden] {name}: loss={loss.item():.6f} trips={K} trips_init={K0} keys={len(out)}")

    # ------------------------------------------------------------------ R10: state_dict manifest (full-size DTU)
    torch.manual_seed(0)
    full = {}
    for dual in (False, True):
        opt = make_opt("DTU", os.path.join(REF, "options/config_hash_sdf.json"), dual, 128)
        sdf = SDF(opt)
        rad = RadF(opt)
        full["dual" if dual else "single"] = {
            "sdf": {k: list(v.shape) for k, v in sdf.state_dict().items()},
            "rad": {k: list(v.shape) for k, v in rad.state_dict().items()},
        }
    manifest["_state_dict_full_dtu"] = full

    # per_level_scale / level geometry the reference requests for each dataset (full-size config)
    geo = {}
    for ds in ("DTU", "ETH3D", "BlendedMVS", "scannet"):
        opt = make_opt(ds, os.path.join(REF, "options/config_hash_sdf.json"), False, 128)
        e = ref_base.get_Embedder(opt=opt, input_dim=3, input_choice="Hash")
        t = e.embedder_obj.table
        geo[ds] = {"per_level_scale": t.per_level_scale, "out_dim": e.out_dim, "n_params": t.n_params}
    manifest["_hash_geometry"] = geo
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("[golden] manifest written")


if __name__ == "__main__":
    main()
