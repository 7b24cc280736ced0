"""the whole optimisation step of the stage loops (ls2fm.stage.RenderStage: sphere tracing -> render with the loss head inside
-> backward -> Adam + ExponentialLR), eager and as ONE captured hipGraph, at the benchmark's shape (C2) and the pipeline's
(C3: DTU, dual field, 8192 rays per step, K <= 10 tracing trips)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ls2fm.numa import bind_to_gpu_numa_node
bind_to_gpu_numa_node(0)          # (the GPU's NUMA node, before the runtime starts: ls2fm/numa.py)
import torch
import bench
from ls2fm import stage
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
dev = "cuda"
for name, ds, rays, n in (("C2", "ETH3D", 1024, 128), ("C3", "DTU", 8192, 128)):
    for capture in (False, True):
        opt = make_options(ds, device=dev, dual_field=True, sample_intvs=n)
        torch.manual_seed(0)
        sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
        bench.randomize([sdf, rad])
        center, ray = bench.synthetic_rays(rays, float(opt.data.bound_max[0]), dev)
        gt = torch.rand(1, rays, 3, device=dev)
        st = stage.RenderStage(opt, ren, sdf, rad, weights=dict(rgb=3, eikonal_loss=2, DC_Loss=0), lr=1e-3, lr_end=1e-4, max_iter=1000,
                               capture=capture)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(10):
                st.step(center, ray, gt)
            torch.cuda.synchronize()
            k = 100
            t = time.perf_counter()
            for _ in range(k):
                st.step(center, ray, gt)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / k
        print(f"{name} ({ds}, {rays} rays x {n}, dual, iters_max {sdf.iters_max}): {'one hipGraph per step' if capture else 'eager':22s} "
              f"{dt * 1e3:7.3f} ms/step = {rays / dt / 1e6:.2f} M rays/s  (trace + render + loss + backward + Adam + schedule)", flush=True)
