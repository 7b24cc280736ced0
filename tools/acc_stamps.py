"""per-workgroup phase times of slab_accumulate from a -DLS2FM_STAMPS build:
   LS2FM_LIB=tools/ab/lib_stamps.so LS2FM_SERIAL=1 python tools/acc_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import bench
from ls2fm import _lib
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer

opt = make_options("ETH3D", device="cuda", dual_field=True, sample_intvs=128)
sdf, rad, ren = SDF(opt).to("cuda"), RadF(opt).to("cuda"), Renderer(opt)
bench.randomize([sdf, rad])
center, ray = bench.synthetic_rays(1024, 5.0, "cuda")
for _ in range(5):
    sdf.zero_grad(); rad.zero_grad()
    bench.loss_head(ren.forward(opt, center, ray, sdf, rad)).backward()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * (8 * 4096))()
fn = lib.ls2fm_debug_acc_stamps
assert fn(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print("workgroups", len(a), "kernel span us", (a[:, 3].max() - t0) / 100.0)
ph = ["start->zeroed+meta", "stream", "flush"]
for lvl in sorted(set(a[:, 4])):
    m = a[a[:, 4] == lvl]
    extra = f" [count/start known {((m[:,6]-m[:,0]).mean())/100.0:5.2f}, LDS zeroed {((m[:,7]-m[:,0]).mean())/100.0:5.2f}]" if (m[:, 6] > 0).all() else ""
    print(f"level {lvl:2d}: wgs {len(m):4d} items/wg {m[:,5].mean():8.0f}  " + extra +
          "  ".join(f"{ph[q]} {((m[:, q + 1] - m[:, q]).mean()) / 100.0:6.2f}" for q in range(3)) +
          f"  total {((m[:, 3] - m[:, 0]).mean()) / 100.0:6.2f} us   starts {((m[:,0].min()-t0)/100.0):6.1f}..{((m[:,0].max()-t0)/100.0):6.1f}")
tot = (a[:, 3] - a[:, 0]).sum() / 100.0
print("sum of workgroup times us", tot, " / 256 CUs =", tot / 256)
# occupancy over time: how many workgroups are alive at each 1-us tick
ticks = np.arange(0, (a[:, 3].max() - t0) / 100.0, 5.0)
alive = [int(((a[:, 0] - t0) / 100.0 <= t).sum() - ((a[:, 3] - t0) / 100.0 <= t).sum()) for t in ticks]
print("alive workgroups every 5 us:", alive)
