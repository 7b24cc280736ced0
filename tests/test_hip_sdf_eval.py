"""GPU parity of the fused no-graph SDF evaluation (ls2fm_sdf_eval) and of the sphere-tracing kernel
(ls2fm_sphere_trace) against the reference golden vectors / the general composed form."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden, rel_err
from helpers import product_for
from ls2fm import fused

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_sdf_eval_vs_reference_golden(case, manifest):
    g = load_golden(case)
    opt, sdf, rad, ren = product_for(manifest[case], g, DEV)
    pts = torch.from_numpy(g["pts"]).to(DEV)
    with torch.no_grad():
        assert fused.can_eval_without_graph(sdf, pts)
        y, feat = sdf.infer_sdf(pts, mode="ret_all")          # dispatches to the fused kernel
        y2 = sdf.infer_sdf(pts.view(5, 8, 3), mode="ret_sdf")
    assert tuple(y.shape) == (40, 1) and tuple(feat.shape) == (40, 17) and tuple(y2.shape) == (5, 8, 1)
    assert rel_err(y.cpu(), g["pts_sdf"]) < 2e-5 and rel_err(feat.cpu(), g["pts_feat"]) < 2e-5
    assert torch.equal(y2.view(-1, 1), y)
    _, _, nrm = fused.sdf_eval(sdf, pts, want_feat=True, want_normal=True)
    assert rel_err(nrm.cpu(), g["pts_normal"]) < 2e-5
    # with a graph requested the composed form is used and must agree with the fused one
    y_graph = sdf.infer_sdf(pts.clone(), mode="ret_sdf")
    assert y_graph.requires_grad and rel_err(y_graph.detach().cpu(), y.cpu()) < 1e-5


def test_sdf_eval_sizes():
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    opt = make_options("DTU", device=DEV)
    sdf = SDF(opt).to(DEV)
    with torch.no_grad():
        assert sdf.infer_sdf(torch.zeros(0, 3, device=DEV)).shape == (0, 1)
        for n in (1, 63, 257, 100000):
            p = torch.rand(n, 3, device=DEV) * 2 - 1
            y = sdf.infer_sdf(p)
            with torch.enable_grad():                      # graph requested -> composed form (HIP grid op + torch MLP)
                y_ref = sdf.infer_sdf(p.clone().requires_grad_(True))
            assert y.shape == (n, 1) and rel_err(y.cpu(), y_ref.detach().cpu()) < 1e-5


@pytest.mark.parametrize("case", ["dtu_single", "dtu_bgsdf"])
def test_sphere_trace_kernel_equals_torch_loop(case, manifest):
    """same trip count, track and far-end distances as the torch-op loop on a well-conditioned field"""
    g = load_golden(case)
    opt, sdf, rad, ren = product_for(manifest[case], g, DEV, sdf_prefix="sdf_init")
    sdf.iters_max = 40
    o = torch.from_numpy(g["st0_center"]).to(DEV)
    d = torch.from_numpy(g["st0_ray"]).to(DEV)
    with torch.no_grad():
        near, far, pts, t_end, k = fused.sphere_trace(sdf, o, d)
        from ls2fm.utils.custom_functions import RayAABBIntersector
        _, hits, _ = RayAABBIntersector.apply(o, d, sdf.center.view(1, 3), sdf.half_size.view(1, 3), 1)
        pts_t, t_end_t, k_t = sdf._trace_loop_torch(o, d, hits[:, 0, 0], hits[:, 0, 1])
    assert k == k_t
    assert torch.equal(near, hits[:, 0, 0]) and torch.equal(far, hits[:, 0, 1])
    fin = torch.isfinite(pts_t)
    assert torch.equal(fin, torch.isfinite(pts))
    assert torch.allclose(pts[fin], pts_t[fin], rtol=1e-4, atol=1e-4)
    fin = torch.isfinite(t_end_t)
    assert torch.allclose(t_end[fin], t_end_t[fin], rtol=1e-4, atol=1e-4)


def test_sphere_trace_wide_and_narrow_kernels_agree(manifest):
    """up to 20 000 rays the tracing kernel spends 16 lanes per ray end, above one lane: the same loop, and the sdf row is
    summed in the same order -> BIT-identical.  50 000 rays in one call (narrow) against the same rays in four calls of
    12 500 (wide)."""
    g = load_golden("dtu_single")
    opt, sdf, rad, ren = product_for(manifest["dtu_single"], g, DEV, sdf_prefix="sdf_init")
    gen = torch.Generator().manual_seed(3)
    n = 50000
    o = (torch.tensor([0.0, 0.0, -2.5]).repeat(n, 1) + 0.05 * torch.randn(n, 3, generator=gen)).to(DEV)
    d = (torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1) + 0.2 * torch.randn(n, 3, generator=gen)).to(DEV)
    cuts = [(q * n // 4, (q + 1) * n // 4) for q in range(4)]
    with torch.no_grad():
        near, far, pts, t_end, k = fused.sphere_trace(sdf, o, d)
        parts = [fused.sphere_trace(sdf, o[a:b], d[a:b]) for a, b in cuts]
    assert k == max(h[4] for h in parts) and k >= 1
    for (a, b), (near_h, far_h, pts_h, t_end_h, k_h) in zip(cuts, parts):
        assert torch.equal(near[a:b], near_h) and torch.equal(far[a:b], far_h)
        kk = min(k, k_h)
        x, y = pts[a:b, :kk], pts_h[:, :kk]
        assert torch.equal(torch.isnan(x), torch.isnan(y)) and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))
        if k_h == k:
            assert torch.equal(torch.nan_to_num(t_end[a:b]), torch.nan_to_num(t_end_h))


def test_infer_sdf_small_and_large_calls_are_bit_identical():
    """sdf-only calls of up to 65 536 points take the 16-lanes-per-point kernel: a point evaluates to the same bits there and
    inside a larger call"""
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    opt = make_options("ETH3D", device=DEV)
    sdf = SDF(opt).to(DEV)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        sdf.embed_fn.embedder_obj.params.copy_(((torch.rand(sdf.embed_fn.embedder_obj.params.shape, generator=gen) * 2 - 1) * 0.1).to(DEV))
        w = sdf.SDF_MLP.mlp[0].weight_v
        w[:, 3:] = (torch.randn(w[:, 3:].shape, generator=gen) * 0.05).to(DEV)
        p = ((torch.rand(200000, 3, generator=gen) * 2 - 1) * 5).to(DEV)
        big = sdf.infer_sdf(p)                       # thread-per-point
        for a, b in ((0, 1), (7, 1000), (1000, 66536), (150000, 200000)):
            assert torch.equal(sdf.infer_sdf(p[a:b].contiguous()), big[a:b]), (a, b)


@pytest.mark.parametrize("dataset,bg", [("ETH3D", False), ("DTU", True), ("scannet", False)])
def test_point_query_forward_small_and_large_calls_are_bit_identical(dataset, bg, monkeypatch):
    """sdf + features + normal of up to 32 768 points take the 16-lanes-per-point kernel (the stage loops' point queries): every
    output of a point has the same bits there and inside a thread-per-point call (forced: LS2FM_POINTS_KERNEL) of a larger
    batch -- also with the background sphere"""
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    opt = make_options(dataset, device=DEV)
    if bg:
        opt.data.bg_sdf = True
    sdf = SDF(opt).to(DEV)
    gen = torch.Generator().manual_seed(9)
    s = float(opt.data.bound_max[0])
    with torch.no_grad():
        sdf.embed_fn.embedder_obj.params.copy_(((torch.rand(sdf.embed_fn.embedder_obj.params.shape, generator=gen) * 2 - 1) * 0.1).to(DEV))
        w = sdf.SDF_MLP.mlp[0].weight_v
        w[:, 3:] = (torch.randn(w[:, 3:].shape, generator=gen) * 0.05).to(DEV)
        p = ((torch.rand(40000, 3, generator=gen) * 2 - 1) * s).to(DEV)
        p[:64] *= 1.3                                  # some points outside the box (and beyond the background sphere)
        monkeypatch.setenv("LS2FM_POINTS_KERNEL", "1")
        big = fused.sdf_eval(sdf, p, want_feat=True, want_normal=True)                 # 40 000 points: thread per point
        big_f = fused.sdf_eval(sdf, p, want_feat=True)
        monkeypatch.delenv("LS2FM_POINTS_KERNEL")
        wide = fused.sdf_eval(sdf, p, want_feat=True, want_normal=True)                # the same 40 000 points, 16 lanes each
        for name, x, y in zip(("sdf", "feat", "normal"), wide, big):
            assert torch.equal(x, y), (name, int((x != y).sum()))
        for a, b in ((0, 1), (0, 17), (5, 5000), (23616, 40000)):
            small = fused.sdf_eval(sdf, p[a:b].contiguous(), want_feat=True, want_normal=True)
            for name, x, y in zip(("sdf", "feat", "normal"), small, big):
                assert torch.equal(x, y[a:b]), (name, a, b, int((x != y[a:b]).sum()))
            small_f = fused.sdf_eval(sdf, p[a:b].contiguous(), want_feat=True)
            assert torch.equal(small_f[0], big_f[0][a:b]) and torch.equal(small_f[1], big_f[1][a:b]), (a, b)
