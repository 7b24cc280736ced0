"""fused point queries (ls2fm_sdf_eval + ls2fm_sdf_points_bwd) against the composed form: one Registration-style iteration
= infer_sdf + gradient (eikonal-type loss) + get_surface_pts on M points, forward + backward.  python tools/time_points.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ls2fm.numa import bind_to_gpu_numa_node
bind_to_gpu_numa_node(0)          # (the GPU's NUMA node, before the runtime starts: ls2fm/numa.py)
import torch
from bench import randomize
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF

dev = "cuda"
opt = make_options("DTU", device=dev)
torch.manual_seed(0)
sdf = SDF(opt).to(dev)
randomize([sdf])
for m in (1024, 8192, 65536):
    pts = (torch.rand(m, 3, device=dev) * 2 - 1)
    row = {}
    for mode in ("fused", "composed"):
        sdf.point_queries = mode

        def it():
            sdf.zero_grad(set_to_none=True)
            p = pts.clone().requires_grad_(True)
            y = sdf.infer_sdf(p)
            n = sdf.gradient(p)
            surf, nlen = sdf.get_surface_pts(p)
            (((n.norm(dim=-1) - 1) ** 2).mean() + y.abs().mean() + (surf ** 2).mean()).backward()
        for _ in range(5):
            it()
        torch.cuda.synchronize()
        t = time.perf_counter()
        k = 30
        for _ in range(k):
            it()
        torch.cuda.synchronize()
        row[mode] = (time.perf_counter() - t) / k * 1e3
    print(f"M={m:6d}: fused {row['fused']:.3f} ms/iteration, composed {row['composed']:.3f} ms/iteration, x{row['composed'] / row['fused']:.1f}", flush=True)
