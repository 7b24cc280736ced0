#!/usr/bin/env python
"""Full-size golden ("one full-size config with only checksums / sparse samples"): the REFERENCE's own
Renderer / SDF / RadF at the shipped L16 / F2 / T19 hash configuration (options/config_hash_sdf.json), DTU bounds, dual
field, 128 samples per ray.  Runs only in the build container (imports /root/reference exactly as make_golden.py does,
same stubs).  Usage:  python tests/golden/make_golden_fullsize.py   -> tests/golden/fullsize_dtu_dual.npz

The two 12 196 240-parameter tables are not stored: they are regenerated from a seed on the CPU generator
(`fullsize_tables` below, shared with the tests) and pinned by a sha256 of their bytes.  Stored: the small weights in
full, the rays, every output of Renderer.forward for them, the gradient of tests/golden/losses.render_loss to every small
parameter and to the rays in full, and for each table gradient its number of non-zeros, fp64 sum, fp64 sum of absolute
values, an fp64 dot product with a seeded probe vector and 1536 sparse samples (the 512 largest entries + 1024 seeded
positions).  Data only; no reference source text.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

TABLE_SEEDS = {"sdf": 9001, "rad": 9002}
PROBE_SEED = 9100
N_RAYS = 16
N_SAMPLES = 128


def fullsize_tables(n_params, amp=0.1):
    """the two hash tables of the full-size golden, U(-amp, amp) from fixed seeds (CPU generator: reproducible)"""
    out = {}
    for name, seed in TABLE_SEEDS.items():
        gen = torch.Generator().manual_seed(seed)
        out[name] = ((torch.rand(n_params, generator=gen) * 2 - 1) * amp).float()
    return out


def probe_vector(n_params):
    return torch.randn(n_params, generator=torch.Generator().manual_seed(PROBE_SEED), dtype=torch.float64)


def sample_positions(n_params, grad):
    top = torch.topk(grad.abs(), 512).indices
    rnd = torch.randint(0, n_params, (1024,), generator=torch.Generator().manual_seed(PROBE_SEED + 1))
    return torch.cat([top, rnd])


def table_digest(grad, prefix, out):
    g64 = grad.double()
    pos = sample_positions(grad.numel(), grad)
    out[prefix + "/nnz"] = np.int64((grad != 0).sum().item())
    out[prefix + "/sum"] = np.float64(g64.sum().item())
    out[prefix + "/abs_sum"] = np.float64(g64.abs().sum().item())
    out[prefix + "/probe_dot"] = np.float64((g64 * probe_vector(grad.numel())).sum().item())
    out[prefix + "/sample_pos"] = pos.numpy().astype(np.int64)
    out[prefix + "/sample_val"] = grad[pos].numpy().copy()


def main():
    import make_golden as MG
    import losses
    assert os.path.isdir(MG.REF), "golden vectors can only be generated where /root/reference exists"
    MG.install_stubs()
    sys.path.insert(0, MG.REF)
    os.chdir(MG.REF)
    import warnings
    warnings.filterwarnings("ignore")
    from models.SDF import SDF
    from models.RadF import RadF
    from models.Renderer import Renderer

    torch.manual_seed(4242)
    gen = torch.Generator().manual_seed(4243)
    opt = MG.make_opt("DTU", os.path.join(MG.REF, "options/config_hash_sdf.json"), True, N_SAMPLES)
    sdf, rad, ren = SDF(opt), RadF(opt), Renderer(opt)
    MG.randomize_module(sdf, gen)
    MG.randomize_module(rad, gen)
    n_params = sdf.embed_fn.embedder_obj.params.numel()
    tabs = fullsize_tables(n_params)
    with torch.no_grad():
        sdf.embed_fn.embedder_obj.params.copy_(tabs["sdf"])
        rad.embed_fn.embedder_obj.params.copy_(tabs["rad"])
    out = {"n_params": np.int64(n_params)}
    for name, t in tabs.items():
        out[f"table_sha256/{name}"] = np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8).copy()
    for mod, pre in ((sdf, "sdf"), (rad, "rad")):
        for k, v in mod.state_dict().items():
            if not k.endswith("embedder_obj.params"):
                out[f"{pre}/{k}"] = v.detach().numpy().copy()
    c, d = MG.make_rays(gen, N_RAYS, 1.0, miss=2, inside=2)
    center = c.view(1, N_RAYS, 3).clone().requires_grad_(True)
    ray = d.view(1, N_RAYS, 3).clone().requires_grad_(True)
    rgb_t = torch.rand(1, N_RAYS, 3, generator=gen)
    nm_dir = torch.randn(3, generator=gen)
    ret = Renderer.forward(ren, opt=opt, center=center, ray=ray, SDF_Field=sdf, Rad_Field=rad)
    loss = losses.render_loss(ret, rgb_t, nm_dir)
    loss.backward()
    out.update({"center": center.detach().numpy(), "ray": ray.detach().numpy(), "rgb_target": rgb_t.numpy(),
                "nm_dir": nm_dir.numpy(), "render_loss": np.float32(loss.item()),
                "d_center": center.grad.numpy().copy(), "d_ray": ray.grad.numpy().copy()})
    for k, v in ret.items():
        out[f"ret/{k}"] = v.detach().numpy()
    for mod, pre in ((sdf, "sdf"), (rad, "rad")):
        for k, p in mod.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            if k.endswith("embedder_obj.params"):
                table_digest(g, f"table_grad/{pre}", out)
            else:
                out[f"render_grad/{pre}/{k}"] = g.numpy().copy()
    # hash indices of a few of the sample points at the finest levels (uint32, bit-exact bar), through the encoder the
    # reference instantiated (oracle stand-in for tcnn: parity unpinned for that third-party part, see oracle/__init__.py)
    np.savez_compressed(os.path.join(HERE, "fullsize_dtu_dual.npz"), **out)
    print(f"[golden] fullsize_dtu_dual: loss={loss.item():.6f} n_params={n_params} "
          f"nnz(sdf)={int(out['table_grad/sdf/nnz'])} nnz(rad)={int(out['table_grad/rad/nnz'])}")


if __name__ == "__main__":
    main()
