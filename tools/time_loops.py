"""per-iteration time of the stage LOOPS (ls2fm.stage.RefineLoop / BALoop / InitLoop) at the benchmark's render shape (ETH3D-like
scene box via the DTU preset, 2 views x 512 rays x 128 samples, dual field, 256 tracked key points per view), eager and captured"""
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
from ls2fm.utils import camera as cam
dev = "cuda"
H, W, n_kp, rays = 96, 128, 256, 1024


def scene(seed=0):
    opt = make_options("DTU", device=dev, dual_field=True, sample_intvs=128)
    opt.Res = 128
    torch.manual_seed(seed)
    sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
    bench.randomize([sdf, rad])
    s = float(opt.data.bound_max[0])
    poses = []
    for v in range(2):
        ang = 0.4 + 0.3 * v
        eye = torch.tensor([2.2 * s * torch.sin(torch.tensor(ang)), 0.25 * s, -2.2 * s * torch.cos(torch.tensor(ang))])
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), fwd); right = right / right.norm()
        R = torch.stack([right, torch.linalg.cross(fwd, right), fwd])
        poses.append(torch.cat([R, (-R @ eye)[:, None]], dim=1))
    poses = torch.stack(poses).to(dev)
    intr = torch.tensor([[0.9 * W, 0.0, W / 2], [0.0, 0.9 * W, H / 2], [0.0, 0.0, 1.0]], device=dev)
    images = torch.rand(2, H * W, 3, device=dev)
    kp = [torch.stack([torch.rand(n_kp, device=dev) * (W - 4) + 2, torch.rand(n_kp, device=dev) * (H - 4) + 2], dim=-1) for _ in range(2)]
    ids = [torch.arange(n_kp, device=dev) for _ in range(2)]
    xyzs = (torch.rand(n_kp, 3, device=dev) - 0.5) * s
    return opt, sdf, rad, ren, stage.TrackedViews(poses, intr, images, kp, ids, xyzs, H, W)


def timed(loop, k=60):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(8):
            loop.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(k):
            loop.step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / k


W_REF = dict(eikonal_loss=2, rgb=3, DC_Loss=0, tracing_loss=2, sdf_surf=2)
W_BA = dict(reproj_error=0, eikonal_loss=2, sdf_surf=2, rgb=3, DC_Loss=0, tracing_loss=1)
W_INIT = dict(reproj_error=0, eikonal_loss=2, sdf_surf=2, rgb=3, DC_Loss=0)
ONLY = os.environ.get("LS2FM_LOOPS", "Refine,Init,BA").split(",")       # e.g. LS2FM_LOOPS=Refine for tools/loop_timeline.py
for capture in (False, True):
  if "Refine" in ONLY:
    opt, sdf, rad, ren, views = scene()
    dt = timed(stage.RefineLoop(opt, ren, sdf, rad, views, weights=W_REF, lr_sdf=1e-3, lr_sdf_end=5e-4, lr_color=1e-3, max_iter=500,
                                rand_rays=rays, capture=capture))
    print(f"RefineLoop  {'captured' if capture else 'eager   '} {dt * 1e3:7.3f} ms/iteration", flush=True)
  if "Init" in ONLY:
    opt, sdf, rad, ren, views = scene()
    dt = timed(stage.InitLoop(opt, ren, sdf, rad, views, weights=W_INIT, lr_sdf=1e-3, lr_sdf_end=1e-4, lr_color=1e-2, max_iter=500,
                              rand_rays=rays, capture=capture))
    print(f"InitLoop    {'captured' if capture else 'eager   '} {dt * 1e3:7.3f} ms/iteration", flush=True)
  if "BA" in ONLY:
    opt, sdf, rad, ren, views = scene()
    dt = timed(stage.BALoop(opt, ren, sdf, rad, views, weights=W_BA, lr_sdf=1e-4, lr_sdf_end=5e-5, lr_color=1e-3, lr_pose_r=5e-3, lr_pose_t=1e-2,
                            max_iter=500, rand_rays=rays, capture=capture))
    print(f"BALoop      {'captured' if capture else 'eager   '} {dt * 1e3:7.3f} ms/iteration", flush=True)
