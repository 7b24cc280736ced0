"""how much of a graph-replayed step is launch-to-launch overhead: the benchmark step captured 1, 2 and 4 times per hipGraph
(measurement only -- bench.py replays ONE step per graph)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from ls2fm.graph import CapturedStep
from ls2fm.losses import RenderLossHead
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
dev = "cuda"
opt = make_options("ETH3D", device=dev, dual_field=True, sample_intvs=128)
sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
bench.randomize([sdf, rad])
center, ray = bench.synthetic_rays(1024, 5.0, dev)
head = RenderLossHead(dev, 3.0, 2.0, 0.0)
gt = torch.full((1, 1024, 3), 0.5, device=dev)
dref = torch.zeros(1, 1024, device=dev)
one = torch.ones((), device=dev)
params = list(sdf.parameters()) + list(rad.parameters())
s_main = torch.cuda.Stream()
torch.cuda.set_stream(s_main)


def step():
    for p in params:
        p.grad = None
    ren.forward_with_loss(opt, center, ray, sdf, rad, head, gt, d_points=dref)[1]["all"].backward(gradient=one)


for k in (1, 2, 4):
    def many(k=k):
        for _ in range(k):
            step()
    cap = CapturedStep(many, params, stream=s_main)
    for _ in range(20):
        cap.replay()
    torch.cuda.synchronize()
    n = 400 // k
    t = time.perf_counter()
    for _ in range(n):
        cap.replay()
    torch.cuda.synchronize()
    print(f"{k} step(s) per graph: {(time.perf_counter() - t) / (n * k) * 1e3:.4f} ms/step")
