"""per-unit phase times of the PERSISTENT slab_accumulate from a -DLS2FM_STAMPS build:
   LS2FM_LIB=tools/ab/lib_stamps.so python tools/acc_stamps_p.py"""
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
assert lib.ls2fm_debug_acc_stamps(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print("units", len(a), "kernel span us", (a[:, 6].max() - t0) / 100.0)
names = ["stream", "mid barrier", "flush", "end barrier"]
cols = [(0, 1), (1, 2), (2, 3), (3, 6)]
for lvl in sorted(set(a[:, 4])):
    m = a[a[:, 4] == lvl]
    print(f"level {lvl:2d}: units {len(m):4d} items/unit {m[:,5].mean():8.0f}  " +
          "  ".join(f"{n} {((m[:, e] - m[:, b]).mean()) / 100.0:6.2f}" for n, (b, e) in zip(names, cols)) +
          f"  total {((m[:, 6] - m[:, 0]).mean()) / 100.0:6.2f} us   starts {((m[:,0].min()-t0)/100.0):6.1f}..{((m[:,0].max()-t0)/100.0):6.1f}")
# per workgroup: units, busy time, end time
wg = {}
for row in a:
    wg.setdefault(int(row[7]), []).append(row)
ends = np.array([max(r[6] for r in rows) for rows in wg.values()]) - t0
n_units = np.array([len(rows) for rows in wg.values()])
print("workgroups", len(wg), "units per workgroup min/mean/max", n_units.min(), n_units.mean(), n_units.max(),
      " end times us min/mean/max", ends.min() / 100.0, ends.mean() / 100.0, ends.max() / 100.0)
gaps = []
for rows in wg.values():
    rows = sorted(rows, key=lambda r: r[0])
    for p, q in zip(rows[:-1], rows[1:]):
        gaps.append((q[0] - p[6]) / 100.0)
print("gap between a unit's end barrier and the next unit's start: mean", np.mean(gaps), "max", np.max(gaps))
last = max(wg, key=lambda b: max(r[6] for r in wg[b]))
print("last workgroup", last, "units (level, items, start us, total us):",
      [(int(r[4]), int(r[5]), round((r[0] - t0) / 100.0, 1), round((r[6] - r[0]) / 100.0, 1)) for r in sorted(wg[last], key=lambda r: r[0])])
if os.environ.get("ACC_STAMPS_LONG"):
    for r in a[a[:, 4] <= 2]:
        print("level", int(r[4]), "uid", int(np.where((a == r).all(1))[0][0]), "items", int(r[5]), "start", round((r[0] - t0) / 100.0, 1),
              " ".join(f"{n} {(r[e] - r[b]) / 100.0:6.2f}" for n, (b, e) in zip(names, cols)), "wg", int(r[7]))
if os.environ.get("ACC_STAMPS_SIM"):
    import heapq
    dur = np.zeros(len(a)); items = np.zeros(len(a), dtype=np.int64); lev = np.zeros(len(a), dtype=np.int64)
    rows = np.nonzero(np.array(buf[:], dtype=np.int64).reshape(4096, 8)[:, 0] > 0)[0]
    full = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
    n = rows.max() + 1
    dur = (full[:n, 6] - full[:n, 0]) / 100.0
    G = 256
    def deal(order_of_wg):
        tot = np.zeros(G)
        for u in range(n): tot[order_of_wg(u)] += dur[u]
        return tot
    snake = deal(lambda u: (u % G) if (u // G) % 2 == 0 else G - 1 - (u % G))
    rr = deal(lambda u: u % G)
    print("sum of unit times / G", dur.sum() / G, " round robin max", rr.max(), " snake min/max", snake.min(), snake.max())
    h = [(0.0, w) for w in range(G)]; heapq.heapify(h)
    for u in range(n):                       # list scheduling in uid order (what a zero-latency claim would do)
        t, w = heapq.heappop(h); heapq.heappush(h, (t + dur[u], w))
    print("list scheduling in uid order: max", max(t for t, _ in h), "min", min(t for t, _ in h))
    h = [(0.0, w) for w in range(G)]; heapq.heapify(h)
    for u in np.argsort(-dur):
        t, w = heapq.heappop(h); heapq.heappush(h, (t + dur[u], w))
    print("LPT: max", max(t for t, _ in h))
