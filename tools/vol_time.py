import sys, time, torch
sys.path.insert(0, 'level-s2fm_official_amd'); sys.path.insert(0, 'tests')
from test_hip_sdf_volume import _field
from ls2fm.utils import util
opt, sdf = _field("ETH3D", 5)
bmax, bmin = [float(v) for v in opt.data.bound_max], [float(v) for v in opt.data.bound_min]
for N in (256, 512):
    util.sdf_volume(sdf, 2.0, N, bmax, bmin); torch.cuda.synchronize()
    t = time.perf_counter()
    util.sdf_volume(sdf, 2.0, N, bmax, bmin); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"N={N}: {dt*1e3:.1f} ms  {N**3/dt/1e9:.2f} G points/s")
# the reference's loop shape: 16 k chunks through infer_sdf with host lattice + copies (first 64 chunks)
import numpy as np
xyz = util.lattice_points(2.0, 512, bmax, bmin, first=0, count=64 * 16384)
torch.cuda.synchronize(); t = time.perf_counter()
out = []
with torch.no_grad():
    for i in range(0, xyz.shape[0], 16384):
        out.append(sdf.infer_sdf(torch.from_numpy(xyz[i:i + 16384]).float().cuda()).data.cpu().numpy())
dt = time.perf_counter() - t
print(f"chunked (64 x 16k through infer_sdf + copies): {dt*1e3:.1f} ms -> 512^3 would take {dt * 8192 / 64:.1f} s")
