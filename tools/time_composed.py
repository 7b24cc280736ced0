"""step time of the fused form vs the general (composed) form, with and without pose gradients"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "level-s2fm_official_amd")); sys.path.insert(0, ROOT)
import bench
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
dev = torch.device("cuda", 0)
opt = make_options("ETH3D", device=str(dev), dual_field=True, sample_intvs=128)
sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
bench.randomize([sdf, rad])
center, ray = bench.synthetic_rays(1024, 5.0, dev)
params = list(sdf.parameters()) + list(rad.parameters())
def run(fn, pose, n=20):
    c = center.clone().requires_grad_(pose); r = ray.clone().requires_grad_(pose)
    def step():
        for p in params: p.grad = None
        c.grad = None; r.grad = None
        bench.loss_head(fn(opt, c, r, sdf, rad)).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("fused, no pose grad      %.3f ms" % run(ren.forward, False))
print("composed, no pose grad   %.3f ms" % run(ren.forward_composed, False))
print("fused, with pose grad    %.3f ms" % run(ren.forward, True))
print("composed, with pose grad %.3f ms" % run(ren.forward_composed, True))
print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
