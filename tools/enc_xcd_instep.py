"""per-XCD end times of the gather pass INSIDE a training step (the launch follows a backward: the L2s do not hold the table
slices any more), from a -DLS2FM_STAMPS build:   LS2FM_LIB=tools/ab/lib_stamps.so python tools/enc_xcd_instep.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from ls2fm import _lib
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
lib = _lib.load()


def step():
    for p in params:
        p.grad = None
    loss = ren.forward_with_loss(opt, center, ray, sdf, rad, head, gt, d_points=dref)[1]["all"]
    loss.backward(gradient=one)


for _ in range(5):
    step()
ends, sums = [], [0.0] * 32
for _ in range(6):
    torch.cuda.synchronize()
    assert lib.ls2fm_debug_enc_reset() == 0
    step()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 56)()
    assert lib.ls2fm_debug_enc_ticks(buf) == 0
    t0 = min(buf[40:48])
    ends.append([(v - t0) / 100.0 for v in buf[32:40]])
    for pl in range(32):
        sums[pl] += buf[pl] / 100.0 / 6
avg = [sum(e[x] for e in ends) / len(ends) for x in range(8)]
print("per XCD end (us after the first start, mean of %d steps): %s  max %.1f" % (len(ends), [round(v, 1) for v in avg], max(avg)))
if not any(sums[16:]):                          # one pass over the entry-interleaved table: 16 pass-levels
    ref = sum(sums[10:16]) / 6
    print("summed workgroup durations per level / fine hashed level (interleaved table):")
    print("  ", [round(v / ref, 2) for v in sums[:16]])
    sys.exit(0)
ref = sum(sums[17::2][-6:]) / 6                 # fine hashed levels of the second grid
print("summed workgroup durations per pass-level / fine hashed level of grid 2 (walking order: level 0 grid 1, level 0 grid 2, ...):")
print("  grid 1:", [round(v / ref, 2) for v in sums[0::2]])
print("  grid 2:", [round(v / ref, 2) for v in sums[1::2]])
