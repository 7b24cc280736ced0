"""cProfile of the host side of one bench step (eager): where the Python / launch time goes"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "level-s2fm_official_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from ls2fm.losses import RenderLossHead
dev = torch.device("cuda", 0)
opt = make_options("ETH3D", device=str(dev), dual_field=True, sample_intvs=128)
sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
bench.randomize([sdf, rad])
center, ray = bench.synthetic_rays(1024, 5.0, dev)
params = list(sdf.parameters()) + list(rad.parameters())
head = RenderLossHead(dev)
gt = torch.full((1, 1024, 3), 0.5, device=dev); dref = torch.zeros(1, 1024, device=dev)
def step():
    for p in params: p.grad = None
    ret = ren.forward(opt, center, ray, sdf, rad)
    head.terms(ret, gt, d_points=dref)[1].backward()
for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per step us", (t1 - t0) / 200 * 1e6, " drain us", (t2 - t1) * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
