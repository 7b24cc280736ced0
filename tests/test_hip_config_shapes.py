"""GPU parity at the FULL shapes of BASELINE.json's configurations 3, 4 and 5, full L16/F2/T19 grids:

  C3  DTU bounds, dual field, 8192 rays per step split over 8 views, 128 samples, + sphere tracing (iters_max 10)
  C4  BlendedMVS bounds, 8 views x 1024 rays, 128 samples (the shape one BA step shards over 8 GPUs)
  C5  ScanNet bounds, 4096 rays x 256 samples (1 M sample points), eikonal term on

Sizes the CPU oracle cannot render whole in test time, so each is checked through
  * a SPARSE ORACLE SAMPLE: a few rays per view rendered by the oracle; the fused outputs of those rays, and the
    gradients of a loss restricted to them (zero cotangent elsewhere -- the backward still processes the whole batch),
    against the oracle to the path's bars (2e-5 outputs, 1e-4 every parameter gradient, d beta vs its exactly summed value);
  * SIZE-INDEPENDENT PROPERTIES: finite outputs, missed rays render the background exactly, depth within [near, far],
    and shard additivity -- the gradient of a sum-type loss over the whole [B,R] batch equals the sum over the B views
    rendered one by one (what the data-parallel path relies on, ls2fm/dist.py);
  * C3: the tracing loop at 8192 rays re-synchronised against the oracle's loop (bit-exact, see
    tests/test_hip_sphere_trace_parity.py) and its differentiable tail against the oracle.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from helpers import named_grads
from test_hip_fused_render import _beta_ok, _randomized
from ls2fm import fused
from ls2fm.options import make_options
from oracle import fields as OF

pytestmark = pytest.mark.gpu
DEV = "cuda"

CONFIGS = {
    # name: dataset, dual, views, rays per view, samples, trace
    "C3_dtu_dual_8192rays": ("DTU", True, 8, 1024, 128, True),
    "C4_bmvs_8views": ("BlendedMVS", False, 8, 1024, 128, False),
    "C5_scannet_4096x256": ("scannet", False, 1, 4096, 256, False),
}


def _view_rays(b, r, s, seed):
    """b cameras on a ring around the box looking at its centre, r unnormalised rays each (utils/camera.py:246-251
    shape), a few rays per view missing the box"""
    gen = torch.Generator().manual_seed(seed)
    cs, ds = [], []
    for v in range(b):
        ang = 2 * np.pi * v / max(b, 1) + 0.3
        eye = torch.tensor([2.5 * s * np.sin(ang), 0.3 * s * np.cos(2 * ang), -2.5 * s * np.cos(ang)], dtype=torch.float32)
        fwd = -eye / eye.norm()
        d = fwd[None, :] + 0.15 * torch.randn(r, 3, generator=gen)
        d[:2] = torch.tensor([0.0, 1.0, 0.0]) + 0.6 * fwd           # glancing / missing rays
        cs.append(eye.repeat(r, 1))
        ds.append(d)
    return torch.stack(cs).to(DEV).contiguous(), torch.stack(ds).float().to(DEV).contiguous()


def _loss(ret, tgt, sel=None, eik_w=0.1):
    """sum-type loss (additive over rays); sel: bool [B,R] restricting it to some rays"""
    w = 1.0 if sel is None else sel[..., None].float()
    rgb = ((ret["rgb"] - tgt).abs() * w).sum()
    eik = (((ret["normals"].norm(dim=-1) - 1.0) ** 2) * w).sum()
    dep = (ret["depth_mlp"] * w).sum()
    nm = ((ret["normal_mlp"] * torch.tensor([0.3, -0.2, 0.5], device=tgt.device, dtype=tgt.dtype)) * w).sum()
    vol = ((ret["sdfs_volume"][..., 0] ** 2) * w).sum()
    return rgb + eik_w * eik + 0.01 * dep + 0.05 * nm + 0.05 * vol


def _all_grads(sdf, rad):
    return {**{"s." + k: v for k, v in named_grads(sdf).items()}, **{"r." + k: v for k, v in named_grads(rad).items()}}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_shape_sparse_oracle_sample_and_properties(name):
    ds, dual, b, r, n, trace = CONFIGS[name]
    opt = make_options(ds, device=DEV, dual_field=dual, sample_intvs=n)
    sdf, rad, ren = _randomized(opt, 51)
    s = float(opt.data.bound_max[0])
    center, ray = _view_rays(b, r, s, 52)
    tgt = torch.rand(b, r, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(53))
    assert fused.can_render(ren, opt, center, ray, sdf, rad)

    # ---- whole batch in one call
    ret = ren.forward(opt, center, ray, sdf, rad)
    assert tuple(ret["rgb"].shape) == (b, r, 3) and tuple(ret["sdfs_volume"].shape) == (b, r, n, 1)
    assert tuple(ret["normals"].shape) == (b, r, n, 3) and tuple(ret["depth_mlp"].shape) == (b, r, 1)
    for k, v in ret.items():
        assert torch.isfinite(v).all(), k
    from ls2fm.ops import ray_aabb_intersect
    cnt, t, _ = ray_aabb_intersect(center.view(-1, 3), ray.view(-1, 3), ren.center.view(1, 3), ren.half_size.view(1, 3), 1)
    near, far = t[:, 0, 0].view(b, r), t[:, 0, 1].view(b, r)
    miss = (cnt == 0).view(b, r)
    assert int(miss.sum()) >= b                                            # the glancing rays
    bg = torch.tensor(opt.data.bgcolor, device=DEV, dtype=torch.float32)
    assert torch.equal(ret["rgb"][miss], bg.expand(int(miss.sum()), 3))  # misses: rgb == bgcolor exactly
    hit = ~miss
    dm = ret["depth_mlp"][..., 0]
    assert (dm[hit] >= near[hit] - 1e-4 * s).all() and (dm[hit] <= far[hit] + 1e-4 * s).all()

    # ---- shard additivity: whole batch vs the views one by one
    sdf.zero_grad(); rad.zero_grad()
    _loss(ret, tgt).backward()
    full = _all_grads(sdf, rad)
    acc = {k: torch.zeros_like(torch.as_tensor(v), dtype=torch.float64) for k, v in full.items()}
    shards = b if b > 1 else 4
    cf, rf, tf = center.reshape(shards, -1, 3), ray.reshape(shards, -1, 3), tgt.reshape(shards, -1, 3)
    for q in range(shards):
        sdf.zero_grad(); rad.zero_grad()
        _loss(ren.forward(opt, cf[q:q + 1].contiguous(), rf[q:q + 1].contiguous(), sdf, rad), tf[q:q + 1]).backward()
        for k, v in _all_grads(sdf, rad).items():
            acc[k] += torch.as_tensor(v).double()
    for k in full:
        assert torch.isfinite(torch.as_tensor(full[k])).all(), k
        assert rel_err(full[k], acc[k]) < (1e-4 if k == "s.beta" else 2e-5), k
    assert torch.as_tensor(full["s.embed_fn.embedder_obj.params"]).abs().max() > 0

    # ---- sparse oracle sample: a few rays per view (one of them a miss), loss restricted to them
    per = max(2, 16 // b)
    gen = torch.Generator().manual_seed(54)
    sel = torch.zeros(b, r, dtype=torch.bool)
    for v in range(b):
        sel[v, torch.randperm(r - 2, generator=gen)[:per] + 2] = True
        sel[v, 0] = True
    sel_d = sel.to(DEV)
    sdf.zero_grad(); rad.zero_grad()
    ret = ren.forward(opt, center, ray, sdf, rad)
    _loss(ret, tgt, sel_d).backward()
    got = _all_grads(sdf, rad)

    cfg = OF.dataset_config(ds, dual_field=dual, sample_intvs=n)
    c_s, r_s, t_s = center[sel_d].cpu().view(1, -1, 3), ray[sel_d].cpu().view(1, -1, 3), tgt[sel_d].cpu().view(1, -1, 3)
    osd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.state_dict().items()}
    ord_ = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in rad.state_dict().items()}
    oret = OF.render(cfg, c_s, r_s, osd, ord_)
    _loss(oret, t_s).backward()
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert rel_err(ret[k][sel_d].cpu(), oret[k][0]) < 2e-5, k
    exact_beta = OF.beta_gradient_exact_sum(cfg, c_s, r_s, {k: v.detach() for k, v in osd.items()},
                                            {k: v.detach() for k, v in ord_.items()},
                                            lambda out: _loss(out, t_s.double()))
    for pre, st in (("s.", osd), ("r.", ord_)):
        for k, v in st.items():
            ref = v.grad if v.grad is not None else torch.zeros_like(v)
            if pre + k == "s.beta":
                _beta_ok(got[pre + k], ref, exact_beta)
                continue
            assert rel_err(got[pre + k], ref) < 1e-4, pre + k

    if not trace:
        return
    # ---- C3: sphere tracing of the whole 8192-ray batch, iters_max = 10 (models/SDF.py:116-226)
    assert sdf.iters_max == cfg.iters_max_st == 10

    def device_field(q):
        with torch.no_grad():
            return sdf.infer_sdf(q.to(DEV).contiguous(), mode="ret_sdf")[:, 0].cpu()
    det = {}
    osd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.state_dict().items()}
    od, os_, _, ofin, otrips = OF.sphere_tracing(cfg, center.cpu(), ray.cpu(), osd, rng=False, loop_field=device_field,
                                                 details=det)
    (od ** 2).sum().backward()
    with torch.no_grad():
        near_t, far_t, pts, t_hist, k = fused.sphere_trace(sdf, center.view(-1, 3), ray.view(-1, 3), history=True)
    assert k == otrips == 10
    assert torch.equal(pts.cpu(), det["track"]) and torch.equal(t_hist.cpu(), det["t_end"])
    sdf.zero_grad()
    d_pred, sdf_last, sampled, finish = sdf.sphere_tracing(center, ray, sdf)
    assert tuple(d_pred.shape) == (b, r) and tuple(finish.shape) == (b * r, 1)
    assert tuple(sampled.shape) == (1, 4096 * k + b * r, 3)
    assert rel_err(d_pred.cpu(), od) < 1e-4 and rel_err(sdf_last.cpu(), os_) < 1e-4
    tie = ((os_.detach().abs() - 2 * s / 10 / cfg.res).abs() < 1e-6).numpy()
    assert np.array_equal(finish.cpu().numpy()[~tie], ofin.numpy()[~tie])
    (d_pred ** 2).sum().backward()
    for name_, v in named_grads(sdf).items():
        ref = osd[name_].grad if osd[name_].grad is not None else torch.zeros_like(osd[name_])
        assert rel_err(v, ref) < 1e-4, name_


def test_maximum_batch_size_matches_its_shards():
    """LS2FM_MAX_RENDER_POINTS = 2^23 sample points in ONE call (64 views x 1024 rays x 128 samples, dual field, full L16/F2/T19 grids):
    the largest batch the ABI accepts, where the kernels' 32-bit element / byte offsets are closest to their range (32 rows x p_pad x 12 B
    = 3.2 GB of Jacobian rows, 16 levels x p_pad x 16 B = 2.1 GB of scatter records).  A ray's outputs do not depend on its batch: the
    whole batch must reproduce its 8 shards BIT FOR BIT, and the gradient of a sum-type loss must be the sum over the shards."""
    b, r, n, shards = 64, 1024, 128, 8
    opt = make_options("DTU", device=DEV, dual_field=True, sample_intvs=n)
    sdf, rad, ren = _randomized(opt, 71)
    s = float(opt.data.bound_max[0])
    center, ray = _view_rays(b, r, s, 72)
    tgt = torch.rand(b, r, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(73))
    assert b * r * n == 1 << 23 and fused.can_render(ren, opt, center, ray, sdf, rad)
    keys = ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp")
    sdf.zero_grad(); rad.zero_grad()
    ret = ren.forward(opt, center, ray, sdf, rad)
    for k in keys:
        assert torch.isfinite(ret[k]).all(), k
    _loss(ret, tgt).backward()
    full = {k: torch.as_tensor(v).detach().clone() for k, v in _all_grads(sdf, rad).items()}
    whole = {k: ret[k].detach().clone() for k in keys}
    del ret
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in full.items()}
    vs = b // shards
    for q in range(shards):
        sdf.zero_grad(); rad.zero_grad()
        sl = slice(q * vs, (q + 1) * vs)
        part = ren.forward(opt, center[sl].contiguous(), ray[sl].contiguous(), sdf, rad)
        for k in keys:
            assert torch.equal(part[k], whole[k][sl]), (k, q)
        _loss(part, tgt[sl]).backward()
        for k, v in _all_grads(sdf, rad).items():
            acc[k] += torch.as_tensor(v).double()
        del part
    for k in full:
        assert torch.isfinite(full[k]).all(), k
        assert rel_err(full[k], acc[k]) < (1e-4 if k == "s.beta" else 2e-5), k
    assert full["s.embed_fn.embedder_obj.params"].abs().max() > 0 and full["r.embed_fn.embedder_obj.params"].abs().max() > 0
    # one ray more is not handed to the fused kernels (the composed form serves it): nothing wraps around
    c1 = torch.cat([center.view(-1, 3), center.view(-1, 3)[:1]]).view(1, -1, 3)
    assert not fused.can_render(ren, opt, c1, c1, sdf, rad)


def test_config1_whole_batch_vs_oracle():
    """BASELINE.json configs[0] at ITS OWN shape -- one synthetic view, 256 rays x 32 samples, single field, full L16/F2/T19 grid,
    the benchmark's synthetic inputs and loss (`bench.py --config C1`): small enough for the CPU oracle to
    render WHOLE, so every output (2e-5) and every parameter gradient (1e-4; d beta against its exactly summed value) of the
    fused HIP path is compared with the oracle on the full batch, not on a sample."""
    import bench
    opt = make_options("DTU", device=DEV, dual_field=False, sample_intvs=32)
    sdf, rad, ren = _randomized(opt, 61)
    s = float(opt.data.bound_max[0])
    center, ray = bench.synthetic_rays(256, s, DEV, seed=0)
    assert fused.can_render(ren, opt, center, ray, sdf, rad)
    sdf.zero_grad(); rad.zero_grad()
    ret = ren.forward(opt, center, ray, sdf, rad)
    bench.loss_head(ret).backward()
    got = _all_grads(sdf, rad)

    cfg = OF.dataset_config("DTU", dual_field=False, sample_intvs=32)
    osd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.state_dict().items()}
    ord_ = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in rad.state_dict().items()}
    oret = OF.render(cfg, center.cpu(), ray.cpu(), osd, ord_)
    bench.loss_head(oret).backward()
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert rel_err(ret[k].cpu(), oret[k]) < 2e-5, k
    exact_beta = OF.beta_gradient_exact_sum(cfg, center.cpu(), ray.cpu(), {k: v.detach() for k, v in osd.items()},
                                            {k: v.detach() for k, v in ord_.items()},
                                            lambda out: bench.loss_head({k: v.double() if torch.is_tensor(v) else v for k, v in out.items()}))
    for pre, st in (("s.", osd), ("r.", ord_)):
        for k, v in st.items():
            ref = v.grad if v.grad is not None else torch.zeros_like(v)
            if pre + k == "s.beta":
                _beta_ok(got[pre + k], ref, exact_beta)
                continue
            assert rel_err(got[pre + k], ref) < 1e-4, pre + k
    assert torch.as_tensor(got["s.embed_fn.embedder_obj.params"]).abs().max() > 0


def test_config2_whole_batch_vs_oracle():
    """The HEADLINE configuration (BASELINE.json configs[1] = `bench.py` default, `--config C2`) compared WHOLE: ETH3D bounds,
    1024 rays x 128 samples, dual field, full L16/F2/T19 grids, the benchmark's synthetic rays and loss head.  The CPU oracle
    renders and differentiates the batch in ~6 s: every output of the fused HIP path (2e-5), every parameter gradient (1e-4;
    d beta against its exactly summed value) and BOTH 12 M-entry table gradients entry by entry (`conftest.per_element_check`:
    every entry above 1e-3 of the largest held to 2e-3 of its own magnitude, zero-ness entry by entry) -- the number on the bench
    line rests on this batch, not on a 48-ray sample of it.  Reference: models/Renderer.py:51-116."""
    import bench
    from conftest import per_element_check
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=128)
    torch.manual_seed(0)
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    bench.randomize([sdf, rad], seed=0)                              # the benchmark's own weights
    s = float(opt.data.bound_max[0])
    center, ray = bench.synthetic_rays(1024, s, DEV, seed=0)
    assert fused.can_render(ren, opt, center, ray, sdf, rad)
    sdf.zero_grad(); rad.zero_grad()
    ret = ren.forward(opt, center, ray, sdf, rad)
    bench.loss_head(ret).backward()
    _check_c2_against_oracle(sdf, rad, center, ray, ret, bench.loss_head,
                             lambda out: bench.loss_head({k: v.double() if torch.is_tensor(v) else v for k, v in out.items()}))


def _check_c2_against_oracle(sdf, rad, center, ray, ret, oracle_loss, oracle_loss_f64, terms=None):
    """outputs, every parameter gradient and both table gradients of the C2 batch against the CPU oracle's render +
    `oracle_loss`; `terms`: (product's loss terms, oracle's term function) compared too"""
    from conftest import per_element_check
    got = _all_grads(sdf, rad)
    cfg = OF.dataset_config("ETH3D", dual_field=True, sample_intvs=128)
    osd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.state_dict().items()}
    ord_ = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in rad.state_dict().items()}
    oret = OF.render(cfg, center.cpu(), ray.cpu(), osd, ord_)
    if terms is not None:
        mine, term_fn = terms
        theirs = term_fn(oret)
        for k in ("rgb_loss", "eikonal_loss", "DC_loss", "mse", "all"):
            assert abs(float(mine[k]) - float(theirs[k])) <= 2e-5 * max(1.0, abs(float(theirs[k]))), (k, float(mine[k]), float(theirs[k]))
    oracle_loss(oret).backward()
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert rel_err(ret[k].cpu(), oret[k]) < 2e-5, k
    exact_beta = OF.beta_gradient_exact_sum(cfg, center.cpu(), ray.cpu(), {k: v.detach() for k, v in osd.items()},
                                            {k: v.detach() for k, v in ord_.items()}, oracle_loss_f64)
    n_tables = 0
    for pre, st in (("s.", osd), ("r.", ord_)):
        for k, v in st.items():
            ref = v.grad if v.grad is not None else torch.zeros_like(v)
            if pre + k == "s.beta":
                _beta_ok(got[pre + k], ref, exact_beta)
                continue
            assert rel_err(got[pre + k], ref) < 1e-4, pre + k
            if k == "embed_fn.embedder_obj.params":
                worst, n_big = per_element_check(got[pre + k], ref, pre + k)
                assert n_big > 1000, (pre + k, n_big)                # the bar is exercised on thousands of entries
                n_tables += 1
    assert n_tables == 2


def test_config2_timed_path_whole_batch_vs_oracle():
    """the path `bench.py` TIMES, compared whole.  Exactly the calls of bench.py's `render_step`
    (`Renderer.forward_with_loss` with the fused `RenderLossHead(w_rgb=3, w_eikonal=2, w_dc=0, global_counts="uniform")`, rgb target
    0.5, traced depth 0, `loss.backward(gradient=one)`; the interleaved table copy trusted as the bench trusts it) on the benchmark's
    own 1024-ray batch and weights: the loss TERMS against oracle/losses.py (pipelines/Camera.py:515-537) on the oracle's render,
    every output (2e-5), every parameter gradient (1e-4; d beta against its exactly summed value) and both 12 M-entry table
    gradients entry by entry -- the same bars as the two-call form above, none edited."""
    import bench
    from oracle.losses import loss_head as oracle_head
    from ls2fm.losses import RenderLossHead
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=128)
    torch.manual_seed(0)
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    bench.randomize([sdf, rad], seed=0)
    n_rays = 1024
    center, ray = bench.synthetic_rays(n_rays, float(opt.data.bound_max[0]), DEV, seed=0)
    assert fused.can_render(ren, opt, center, ray, sdf, rad)
    fused.trust_mirror_in_capture(sdf, rad)
    head = RenderLossHead(DEV, w_rgb=3.0, w_eikonal=2.0, w_dc=0.0, global_counts="uniform")
    rgb_gt = torch.full((1, n_rays, 3), 0.5, device=DEV)
    depth_ref = torch.zeros(1, n_rays, device=DEV)
    one = torch.ones((), device=DEV)
    for p in list(sdf.parameters()) + list(rad.parameters()):
        p.grad = None
    ret, L = ren.forward_with_loss(opt, center, ray, sdf, rad, head, rgb_gt, d_points=depth_ref)
    L["all"].backward(gradient=one)

    def terms(out, f64=False):
        gt = torch.full((1, n_rays, 3), 0.5, dtype=torch.float64 if f64 else torch.float32)
        dp = torch.zeros(1, n_rays, dtype=gt.dtype)
        return oracle_head(out, gt, dp, None, None, None, 3.0, 2.0, 0.0)
    _check_c2_against_oracle(sdf, rad, center, ray, ret, lambda out: terms(out)["all"],
                             lambda out: terms({k: v.double() if torch.is_tensor(v) else v for k, v in out.items()}, True)["all"],
                             terms=(L, terms))
