"""sphere tracing (ls2fm_sphere_trace) and no-graph infer_sdf throughput at the stage-loop sizes (BASELINE config C3: 8192 rays)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from ls2fm import fused
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF

opt = make_options("DTU", device="cuda", dual_field=True)
sdf = SDF(opt).to("cuda")
bench.randomize([sdf])
for R in (1024, 8192, 16384, 24576, 32768, 65536):
    c, d = bench.synthetic_rays(R, 1.0, "cuda")
    o, dd = c.view(-1, 3), d.view(-1, 3)
    with torch.no_grad():
        for _ in range(3):
            out = fused.sphere_trace(sdf, o, dd)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 20
        for _ in range(n):
            out = fused.sphere_trace(sdf, o, dd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"sphere_trace {R:6d} rays (iters_max {sdf.iters_max}, trips {out[4]}): {dt * 1e3:7.3f} ms  {R / dt / 1e6:6.2f} M rays/s")
for n_pts in (1 << 17, 1 << 20, 1 << 24):
    p = (torch.rand(n_pts, 3, device="cuda") * 2 - 1)
    with torch.no_grad():
        sdf.infer_sdf(p); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            sdf.infer_sdf(p)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"infer_sdf {n_pts:9d} random points: {dt * 1e3:8.3f} ms  {n_pts / dt / 1e9:5.2f} G points/s")
# kernel-level split of one sphere_trace call
from ls2fm import _lib, profile as prof
lib = _lib.load()
c, d = bench.synthetic_rays(8192, 1.0, "cuda")
o, dd = c.view(-1, 3), d.view(-1, 3)
lib.ls2fm_profile_reset(); lib.ls2fm_profile_enable(1)
with torch.no_grad():
    for _ in range(20):
        fused.sphere_trace(sdf, o, dd)
torch.cuda.synchronize()
lib.ls2fm_profile_enable(0)
print({k: round(v[0], 1) for k, v in prof.kernel_times(lib).items()})
# small-batch latency of the no-graph SDF evaluation
for n_pts in (1024, 8192, 65536):
    p = (torch.rand(n_pts, 3, device="cuda") * 2 - 1)
    lib.ls2fm_profile_reset(); lib.ls2fm_profile_enable(1)
    with torch.no_grad():
        for _ in range(20):
            fused.sdf_eval(sdf, p, want_feat=True, want_normal=True)
    torch.cuda.synchronize(); lib.ls2fm_profile_enable(0)
    print(n_pts, "points, sdf+feat+normal:", {k: round(v[0], 1) for k, v in prof.kernel_times(lib).items()})
for n_pts in (1024, 8192, 65536, 131072, 262144):
    p = (torch.rand(n_pts, 3, device="cuda") * 2 - 1)
    lib.ls2fm_profile_reset(); lib.ls2fm_profile_enable(1)
    with torch.no_grad():
        for _ in range(20):
            sdf.infer_sdf(p)
    torch.cuda.synchronize(); lib.ls2fm_profile_enable(0)
    print(n_pts, "points, sdf only:", {k: round(v[0], 1) for k, v in prof.kernel_times(lib).items()})
