import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from ls2fm import stage
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
dev = "cuda"
opt = make_options("ETH3D", device=dev, dual_field=True, sample_intvs=128)
torch.manual_seed(0)
sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
bench.randomize([sdf, rad])
center, ray = bench.synthetic_rays(1024, 5.0, dev)
gt = torch.rand(1, 1024, 3, device=dev)
st = stage.RenderStage(opt, ren, sdf, rad, weights=dict(rgb=3, eikonal_loss=2, DC_Loss=0), lr=1e-3, lr_end=1e-4, max_iter=1000, capture=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(30):
        st.step(center, ray, gt)
    torch.cuda.synchronize()
