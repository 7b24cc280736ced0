#!/usr/bin/env python
"""sha256 of every output and gradient of ONE benchmark step (bench.py's own batch, weights and fused loss head): two library
builds (LS2FM_LIB=...) or two settings that must not change a bit can be compared across processes.
    python tools/grad_digest.py [--config C2] [--single-field]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "level-s2fm_official_amd")]
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--single-field", action="store_true")
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
dual = cfg["dual"] and not a.single_field
from ls2fm import fused
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from ls2fm.losses import RenderLossHead
dev = "cuda"
opt = make_options(cfg["dataset"], device=dev, dual_field=dual, sample_intvs=cfg["samples"])
torch.manual_seed(0)
sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
bench.randomize([sdf, rad], seed=0)
center, ray = bench.synthetic_rays(cfg["rays"], float(opt.data.bound_max[0]), dev, seed=0)
head = RenderLossHead(dev, w_rgb=3.0, w_eikonal=2.0, w_dc=0.0, global_counts="uniform")
gt = torch.full((1, cfg["rays"], 3), 0.5, device=dev)
dref = torch.zeros(1, cfg["rays"], device=dev)
h = hashlib.sha256()
for rep in range(2):
    for p in list(sdf.parameters()) + list(rad.parameters()):
        p.grad = None
    ret, L = ren.forward_with_loss(opt, center, ray, sdf, rad, head, gt, d_points=dref)
    L["all"].backward()
    torch.cuda.synchronize()
    for k in sorted(ret):
        if torch.is_tensor(ret[k]):
            h.update(ret[k].detach().cpu().numpy().tobytes())
    for m in (sdf, rad):
        for n, p in m.named_parameters():
            if p.grad is not None:
                h.update(n.encode()); h.update(p.grad.detach().cpu().numpy().tobytes())
print(a.config, "single" if not dual else "dual", h.hexdigest()[:24], "loss", float(L["all"]))
