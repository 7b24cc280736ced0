#!/usr/bin/env python
"""Does `shade_bwd`'s run-time mode (121 vs 133 us at C2, profiles/r06_notes.md section 6) follow the PLACEMENT of the render workspace?
One process; between measurements a pad tensor of varying size is allocated while the workspace is free, so that the caching allocator
hands the next workspace out at another address.  Prints the kernel spans per placement.   python tools/mode_probe.py [--single-field]"""
import os, sys


def _cpus(arg):
    if arg not in sys.argv:
        return None
    a, b = sys.argv[sys.argv.index(arg) + 1].split("-")
    return set(range(int(a), int(b) + 1))


# --init-cpus a-b: CPU affinity while the runtime, the library and every allocation are set up; --run-cpus c-d: affinity while measuring
if _cpus("--init-cpus"):
    os.sched_setaffinity(0, _cpus("--init-cpus"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "level-s2fm_official_amd")]
if "--bind" in sys.argv:
    from ls2fm.numa import bind_to_gpu_numa_node
    print("bound to NUMA node", bind_to_gpu_numa_node(0), "cpus", len(os.sched_getaffinity(0)), flush=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "level-s2fm_official_amd")]
import torch
import bench
from ls2fm import _lib, fused
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from ls2fm.losses import RenderLossHead
from ls2fm.profile import kernel_times

dual = "--single-field" not in sys.argv
dev = "cuda"
lib = _lib.load()
opt = make_options("ETH3D", device=dev, dual_field=dual, sample_intvs=128)
torch.manual_seed(0)
sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
bench.randomize([sdf, rad], seed=0)
center, ray = bench.synthetic_rays(1024, float(opt.data.bound_max[0]), dev, seed=0)
head = RenderLossHead(dev, w_rgb=3.0, w_eikonal=2.0, w_dc=0.0)
gt = torch.full((1, 1024, 3), 0.5, device=dev)
dref = torch.zeros(1, 1024, device=dev)
params = list(sdf.parameters()) + list(rad.parameters())
seen = []
orig_empty = torch.empty


def step():
    for p in params:
        p.grad = None
    ret, L = ren.forward_with_loss(opt, center, ray, sdf, rad, head, gt, d_points=dref)
    L["all"].backward()


def measure(tag):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.ls2fm_profile_reset(); lib.ls2fm_profile_enable(1)
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    lib.ls2fm_profile_enable(0)
    t = kernel_times(lib)
    print(f"trial {tag:28s} " + "  ".join(f"{k} {v[0]:.1f}" for k, v in sorted(t.items(), key=lambda kv: -kv[1][2])[:6]), flush=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
if _cpus("--run-cpus"):
    os.sched_setaffinity(0, _cpus("--run-cpus"))
if "--streams" in sys.argv:
    # the same steps on the default stream and on freshly created streams (HIP maps streams to a few hardware queues round-robin)
    measure("default stream")
    streams = [torch.cuda.Stream() for _ in range(0 if "--quick" in sys.argv else 9)]
    for i, st in enumerate(streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            measure(f"stream {i} (prio 0)")
    hi = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(hi):
        measure("high-priority stream")
    measure("default stream again")
else:
    pads = []
    for trial, pad_mb in enumerate([0, 0, 1, 3, 7, 16, 33, 64, 130, 0, 257, 2]):
        torch.cuda.synchronize()
        if pad_mb:
            pads.append(torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev))      # kept: the workspace's old block is split / displaced
        measure(f"{trial:2d} pad {pad_mb:4d} MB reserved {torch.cuda.memory_reserved() >> 20:6d}")
