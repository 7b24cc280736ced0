"""Fused point queries with a graph (ls2fm_sdf_eval forward + ls2fm_sdf_points_bwd backward behind SDF.infer_sdf / gradient /
get_surface_pts) against the composed form (HIP hash-grid op + torch layers + autograd double backward) and the CPU oracle,
full L16/T19 grid, M up to 10^5, incl. gradients w.r.t. the points (mixed second partials) and ragged / tiny M."""
import pytest
import torch

from conftest import rel_err
from helpers import named_grads
from test_hip_fused_render import _randomized
from ls2fm import fused
from ls2fm.options import make_options

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scalar(sdf, pts, with_feat=True):
    """a scalar touching every output of the three entry points: infer_sdf (sdf + features), gradient (eikonal-type and a
    linear term: double backward), get_surface_pts"""
    p = pts.clone().requires_grad_(True)
    y, feat = sdf.infer_sdf(p, mode="ret_all")
    n = sdf.gradient(p)
    surf, nlen = sdf.get_surface_pts(p)
    w = torch.tensor([0.3, -0.5, 0.8], device=pts.device)
    val = ((n.norm(dim=-1) - 1) ** 2).mean() + 0.2 * (n * w).sum(-1).mean() + 0.1 * y.mean() + 0.3 * (surf ** 2).mean() \
        + 0.05 * nlen.mean()
    if with_feat:
        val = val + 0.05 * (feat[..., 1:] ** 2).mean()
    return val, p, (y, feat, n, surf, nlen)


@pytest.mark.parametrize("ds,m", [("DTU", 1), ("ETH3D", 37), ("BlendedMVS", 4096), ("scannet", 100000)])
def test_fused_point_queries_equal_composed(ds, m):
    opt = make_options(ds, device=DEV)
    sdf, rad, ren = _randomized(opt, 91)
    s = float(opt.data.bound_max[0])
    pts = ((torch.rand(m, 3, generator=torch.Generator().manual_seed(92)) * 2 - 1) * s * 1.02).to(DEV)    # a few outside the box
    assert fused.can_query_points(sdf, pts)
    res = {}
    for mode in ("fused", "composed"):
        sdf.point_queries = mode
        sdf.zero_grad()
        val, p, outs = _scalar(sdf, pts)
        val.backward()
        res[mode] = (val.detach(), p.grad.clone(), named_grads(sdf), [o.detach() for o in outs])
    assert abs(float(res["fused"][0]) - float(res["composed"][0])) < 1e-5 * abs(float(res["composed"][0]))
    for a, b in zip(res["fused"][3], res["composed"][3]):
        assert rel_err(a.cpu(), b.cpu()) < 2e-5
    assert rel_err(res["fused"][1].cpu(), res["composed"][1].cpu()) < 1e-4
    for k, v in res["fused"][2].items():
        assert rel_err(v, res["composed"][2][k]) < 1e-4, k
    assert torch.as_tensor(res["fused"][2]["embed_fn.embedder_obj.params"]).abs().max() > 0


def test_fused_point_queries_vs_cpu_oracle():
    from oracle import fields as OF
    opt = make_options("ETH3D", device=DEV)
    sdf, rad, ren = _randomized(opt, 93)
    pts = ((torch.rand(600, 3, generator=torch.Generator().manual_seed(94)) * 2 - 1) * 5.0).to(DEV)
    sdf.point_queries = "fused"
    val, p, outs = _scalar(sdf, pts)
    val.backward()
    cfg = OF.dataset_config("ETH3D")
    table = cfg.table()
    osd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.state_dict().items()}
    op = pts.cpu().clone().requires_grad_(True)
    y, feat = OF.infer_sdf(op, osd, cfg, table, "ret_all")
    n = OF.sdf_gradient(op, osd, cfg, table)
    surf, nlen = OF.get_surface_pts(op, osd, cfg, table)
    w = torch.tensor([0.3, -0.5, 0.8])
    oval = ((n.norm(dim=-1) - 1) ** 2).mean() + 0.2 * (n * w).sum(-1).mean() + 0.1 * y.mean() + 0.3 * (surf ** 2).mean() \
        + 0.05 * nlen.mean() + 0.05 * (feat[..., 1:] ** 2).mean()
    oval.backward()
    assert abs(float(val) - float(oval)) < 1e-5 * abs(float(oval))
    for a, b in zip(outs, (y, feat, n, surf, nlen)):
        assert rel_err(a.detach().cpu(), b) < 2e-5
    assert rel_err(p.grad.cpu(), op.grad) < 1e-4
    for k, v in named_grads(sdf).items():
        ref = osd[k].grad if osd[k].grad is not None else torch.zeros_like(osd[k])
        assert rel_err(v, ref) < 1e-4, k


def test_point_queries_leave_no_grad_paths_and_shapes_alone():
    opt = make_options("DTU", device=DEV)
    sdf, rad, ren = _randomized(opt, 95)
    pts = torch.rand(5, 7, 3, device=DEV) - 0.5
    with torch.no_grad():
        y0 = sdf.infer_sdf(pts)
    y1, f1 = sdf.infer_sdf(pts.clone().requires_grad_(True), mode="ret_all")
    assert tuple(y1.shape) == (5, 7, 1) and tuple(f1.shape) == (5, 7, 17) and y1.requires_grad and f1.requires_grad
    assert torch.equal(y0, y1.detach())                           # the same forward kernel with and without a graph
    q = pts.clone()
    n = sdf.gradient(q)
    assert q.requires_grad and tuple(n.shape) == (5, 7, 3) and n.requires_grad
    with torch.no_grad():
        n2 = sdf.gradient(pts.clone())                            # the reference enables grad inside (SDF.py:103)
    assert n2.requires_grad and rel_err(n2.detach().cpu(), n.detach().cpu()) < 1e-5


@pytest.mark.parametrize("ds,m", [("ETH3D", 37), ("DTU", 3000), ("scannet", 16384)])
def test_point_query_backward_kernels_are_bit_identical(ds, m, monkeypatch):
    """up to 16 384 points the backward's per-point kernels run 16 lanes per point (the stage loops' sizes), beyond it one
    thread per point: forced onto the SAME points (LS2FM_POINTS_KERNEL) the two leave the same bits in every gradient -- table,
    MLP and the points themselves"""
    opt = make_options(ds, device=DEV)
    sdf, rad, ren = _randomized(opt, 93)
    s = float(opt.data.bound_max[0])
    pts = ((torch.rand(m, 3, generator=torch.Generator().manual_seed(94)) * 2 - 1) * s * 1.02).to(DEV)
    sdf.point_queries = "fused"
    res = {}
    for which in ("1", "2"):
        monkeypatch.setenv("LS2FM_POINTS_KERNEL", which)
        sdf.zero_grad()
        val, p, outs = _scalar(sdf, pts)
        val.backward()
        res[which] = (p.grad.clone(), {k: torch.as_tensor(v).clone() for k, v in named_grads(sdf).items()})
    assert torch.equal(res["1"][0], res["2"][0])
    for k, v in res["1"][1].items():
        assert torch.equal(v, res["2"][1][k]), k
    assert res["1"][1]["embed_fn.embedder_obj.params"].abs().max() > 0


def test_surface_points_of_graphless_points_take_one_query_node():
    """get_surface_pts on points WITHOUT a graph (BA's tracked points) evaluates value and normal in one query node: same
    outputs, and parameter gradients equal to the sum the two separate nodes (points with a graph) leave"""
    opt = make_options("ETH3D", device=DEV)
    sdf, rad, ren = _randomized(opt, 95)
    s = float(opt.data.bound_max[0])
    pts = ((torch.rand(2500, 3, generator=torch.Generator().manual_seed(96)) * 2 - 1) * s).to(DEV)
    sdf.point_queries = "fused"
    res = {}
    for which in ("one", "two"):
        sdf.zero_grad()
        p = pts.clone()
        if which == "two":
            p.requires_grad_(True)                       # the separate-node path
        surf, nlen = sdf.get_surface_pts(p)
        assert p.requires_grad                           # the reference's side effect (SDF.py:104) either way
        again = sdf.infer_sdf(surf, mode="ret_sdf")
        ((surf ** 2).mean() + 0.1 * nlen.mean() + again.abs().mean()).backward()
        res[which] = (surf.detach(), nlen.detach(), named_grads(sdf))
    assert rel_err(res["one"][0].cpu(), res["two"][0].cpu()) < 1e-6 and rel_err(res["one"][1].cpu(), res["two"][1].cpu()) < 1e-6
    for k, v in res["one"][2].items():
        assert rel_err(v, res["two"][2][k]) < 2e-5, k
