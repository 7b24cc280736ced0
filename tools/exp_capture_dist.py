"""experiment: can the sharded step (RCCL collectives on a communication stream, issued from inside the backward) be captured
into ONE hipGraph?  1-rank RCCL group (LS2FM_DIST_SINGLE=1).  usage: python tools/exp_capture_dist.py <n_groups>"""
import os, sys, time
os.environ.setdefault("LS2FM_DIST_SINGLE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import torch.distributed as dist
import bench
from ls2fm import fused
from ls2fm.dist import ShardedAdam
from ls2fm.graph import CapturedStep
from ls2fm.losses import RenderLossHead
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
opt = make_options("ETH3D", device=str(dev), dual_field=True, sample_intvs=128)
torch.manual_seed(0)
sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
bench.randomize([sdf, rad], seed=0)
center, ray = bench.synthetic_rays(1024, 5.0, dev, seed=0)
so = ShardedAdam.for_fields(sdf, rad, lr=1e-4, lr_color=1e-4, scheduled_gamma=1.0, n_groups=n_groups, async_gather=n_groups > 1)
params = list(so.params)
head = RenderLossHead(dev, w_rgb=3.0, w_eikonal=2.0, w_dc=0.0, global_counts="uniform")
rgb_gt = torch.full((1, 1024, 3), 0.5, device=dev)
depth_ref = torch.zeros(1, 1024, device=dev)
one = torch.ones((), device=dev)


def step():
    so.wait_params()
    for p in params:
        p.grad = None
    loss = ren.forward_with_loss(opt, center, ray, sdf, rad, head, rgb_gt, d_points=depth_ref)[1]["all"]
    loss.backward(gradient=one)
    so.step()
    return loss


s_main = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s_main)
for _ in range(20):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(f"n_groups {n_groups}: eager {(time.perf_counter() - t) / 200 * 1e3:.3f} ms/step", flush=True)
try:
    cap = CapturedStep(step, params, stream=s_main)
    for _ in range(20):
        cap.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        cap.replay()
    torch.cuda.synchronize()
    print(f"n_groups {n_groups}: hipGraph {(time.perf_counter() - t) / 200 * 1e3:.3f} ms/step", flush=True)
except Exception as e:                      # noqa: BLE001
    print(f"n_groups {n_groups}: capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
dist.destroy_process_group()
