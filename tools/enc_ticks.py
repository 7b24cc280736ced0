"""summed workgroup durations of the gather pass per (grid, level) from a -DLS2FM_STAMPS build:
   LS2FM_LIB=tools/ab/lib_stamps.so python tools/enc_ticks.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
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
n = 20
lib = _lib.load()
train = len(sys.argv) > 1 and sys.argv[1] == "train"      # grad enabled: the launch also counts the scatter's items
with (torch.enable_grad() if train else torch.no_grad()):
    for _ in range(3):
        ren.forward(opt, center, ray, sdf, rad)
    torch.cuda.synchronize()
    assert lib.ls2fm_debug_enc_reset() == 0
    ren.forward(opt, center, ray, sdf, rad)            # one launch: per-XCD first start / last end
    torch.cuda.synchronize()
    one = (ctypes.c_ulonglong * 56)()
    assert lib.ls2fm_debug_enc_ticks(one) == 0
    t0 = min(one[40:48])
    print("per XCD: start %s  end %s (us after the first start)" % ([round((v - t0) / 100.0, 1) for v in one[40:48]],
                                                                   [round((v - t0) / 100.0, 1) for v in one[32:40]]))
    assert lib.ls2fm_debug_enc_reset() == 0
    for _ in range(n):
        ren.forward(opt, center, ray, sdf, rad)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 56)()
assert lib.ls2fm_debug_enc_ticks(buf) == 0
t = [b / n / 100.0 for b in buf]          # us of summed workgroup time per launch
ref = max(t)
d = sdf.embed_fn.embedder_obj.desc
for pl in range(32):
    l = pl >> 1                                  # walking order: (level 0, grid 1), (level 0, grid 2), (level 1, grid 1), ...
    print(f"grid {pl & 1} level {l:2d} scale {d.scale[l]:8.1f} hashed {d.hashed[l]}  sum {t[pl]:9.1f} us  rel {t[pl] / ref:5.2f}")
