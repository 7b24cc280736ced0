#!/usr/bin/env python
"""host-side cost of one eager benchmark step (cProfile over 300 steps; no device sync inside the loop)"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from ls2fm.losses import RenderLossHead
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
    cfg = bench.CONFIGS["C2"]
    opt = make_options(cfg["dataset"], device=str(dev), dual_field=cfg["dual"], sample_intvs=cfg["samples"])
    torch.manual_seed(0)
    sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
    bench.randomize([sdf, rad], seed=0)
    center, ray = bench.synthetic_rays(cfg["rays"], float(opt.data.bound_max[0]), dev, seed=0)
    params = list(sdf.parameters()) + list(rad.parameters())
    head = RenderLossHead(dev, 3.0, 2.0, 0.0)
    rgb_gt = torch.full((1, cfg["rays"], 3), 0.5, device=dev)
    depth_ref = torch.zeros(1, cfg["rays"], device=dev)
    one = torch.ones((), device=dev)
    s = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(s)
    def step():
        for p in params:
            p.grad = None
        loss = ren.forward_with_loss(opt, center, ray, sdf, rad, head, rgb_gt, d_points=depth_ref)[1]["all"]
        loss.backward(gradient=one)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(300):
        step()
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    tot = time.perf_counter() - t
    print(f"host enqueue {host / 300 * 1e6:.0f} us/step, with device {tot / 300 * 1e6:.0f} us/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        step()
    pr.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(35)
    print(out.getvalue()[:6000])

main()
