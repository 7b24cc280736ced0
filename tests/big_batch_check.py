"""one-off: a 32768-ray x 128-sample batch (4.2 M sample points) in ONE fused call against the sum of 8 shards:
   python tests/big_batch_check.py"""
import sys, torch
sys.path.insert(0, 'tests')
import conftest  # noqa: F401,E402  (sets the import paths)
from test_hip_fused_render import _randomized, _rays
from helpers import named_grads
from conftest import rel_err
from ls2fm.options import make_options
DEV = "cuda"
opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=128)
sdf, rad, ren = _randomized(opt, 31)
R = 32768
center, ray = _rays(R, 5.0, 32)
tgt = torch.rand(1, R, 3, device=DEV)
def loss_of(ret, sl):
    return ((ret["rgb"] - tgt[:, sl]).abs().sum() + 0.1 * ((ret["normals"].norm(dim=-1) - 1.0) ** 2).sum() + 0.01 * ret["depth_mlp"].sum())
sdf.zero_grad(); rad.zero_grad()
torch.cuda.synchronize()
import time; t = time.perf_counter()
loss_of(ren.forward(opt, center, ray, sdf, rad), slice(None)).backward()
torch.cuda.synchronize(); print("one call of", R, "rays:", time.perf_counter() - t, "s; peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
full = {**named_grads(sdf), **{"r." + k: v for k, v in named_grads(rad).items()}}
acc = {k: torch.zeros_like(torch.as_tensor(v), dtype=torch.float64) for k, v in full.items()}
for q in range(8):
    sl = slice(R // 8 * q, R // 8 * (q + 1))
    sdf.zero_grad(); rad.zero_grad()
    loss_of(ren.forward(opt, center[:, sl].contiguous(), ray[:, sl].contiguous(), sdf, rad), sl).backward()
    for k, v in {**named_grads(sdf), **{"r." + k: v for k, v in named_grads(rad).items()}}.items():
        acc[k] += torch.as_tensor(v).double()
worst = max((rel_err(full[k], acc[k]), k) for k in full)
print("worst rel err full vs 8 shards:", worst)
