"""The loss head evaluated INSIDE the render (ls2fm_render_opts.loss; Renderer.forward_with_loss): partial sums in the
forward's last kernel, the upstream of rgb / normals / depth formed in the backward's first.  Against (a) the two-call form
(Renderer.forward, then RenderLossHead as its own kernels), (b) the CPU oracle's render + the torch restatement of the
reference's loss expressions (oracle/losses.py: pipelines/Camera.py:515-537, BA.py:193-218), (c) sharded evaluation with
global counts = the single-process gradients (what a multi-GPU run all-reduces)."""
import pytest
import torch

from conftest import rel_err
from helpers import named_grads
from test_hip_fused_render import _randomized, _rays
from ls2fm.losses import RenderLossHead
from ls2fm.options import make_options

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _grads(sdf, rad):
    return {**{"s." + k: v for k, v in named_grads(sdf).items()}, **{"r." + k: v for k, v in named_grads(rad).items()}}


def _setup(ds, dual, n_samples, n_rays, seed, hash_encoding=None):
    opt = make_options(ds, device=DEV, dual_field=dual, sample_intvs=n_samples, hash_encoding=hash_encoding)
    sdf, rad, ren = _randomized(opt, seed)
    center, ray = _rays(n_rays, float(opt.data.bound_max[0]), seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    gt = torch.rand(1, n_rays, 3, generator=g).to(DEV)
    d_points = (torch.rand(1, n_rays, generator=g) * 3.0 * float(opt.data.bound_max[0])).to(DEV)
    masks = {k: (torch.rand(1, n_rays, generator=g) < p).to(DEV) for k, p in (("finish", 0.5), ("eik", 0.7), ("bg", 0.8))}
    return opt, sdf, rad, ren, center, ray, gt, d_points, masks


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("ds,dual,n_samples,n_rays", [("ETH3D", True, 64, 96), ("DTU", False, 33, 50), ("scannet", False, 300, 9)])
def test_fused_loss_equals_two_call_form(ds, dual, n_samples, n_rays, masked):
    opt, sdf, rad, ren, center, ray, gt, d_points, masks = _setup(ds, dual, n_samples, n_rays, 61)
    head = RenderLossHead(DEV, 3.0, 2.0, 1.0)
    kw = dict(mask_finish=masks["finish"], mask_eik=masks["eik"], mask_bg=masks["bg"]) if masked else {}
    res = {}
    for form in ("fused", "split"):
        c, r = center.clone().requires_grad_(True), ray.clone().requires_grad_(True)
        dp = d_points.clone().requires_grad_(True)
        sdf.zero_grad(); rad.zero_grad()
        if form == "fused":
            ret, L = ren.forward_with_loss(opt, c, r, sdf, rad, head, gt, d_points=dp, **kw)
        else:
            ret = ren.forward(opt, c, r, sdf, rad)
            L = head(ret, gt, d_points=dp, **kw)
        # the weighted total, individual terms and an explicit use of an output, all at once
        (L["all"] + 0.3 * L["mse"] + 2.0 * L["eikonal_loss"] + 0.05 * ret["depth_mlp"].sum() + 0.01 * (ret["rgb"] ** 2).sum()).backward()
        res[form] = (ret, L, _grads(sdf, rad), c.grad.clone(), r.grad.clone(), dp.grad.clone())
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert torch.equal(res["fused"][0][k], res["split"][0][k]), k
    for k in ("rgb_loss", "eikonal_loss", "DC_loss", "mse", "all"):
        a, b = float(res["fused"][1][k]), float(res["split"][1][k])
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (k, a, b)
    for k, v in res["fused"][2].items():
        assert rel_err(v, res["split"][2][k]) < 2e-5, k
    for q in (3, 4, 5):
        assert rel_err(res["fused"][q].cpu(), res["split"][q].cpu()) < 2e-5


def test_fused_loss_vs_cpu_oracle():
    from oracle import fields as OF
    from oracle.losses import loss_head
    enc = dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16)
    opt, sdf, rad, ren, center, ray, gt, d_points, masks = _setup("BlendedMVS", True, 24, 40, 71, hash_encoding=enc)
    head = RenderLossHead(DEV, 3.0, 2.0, 0.5)
    dp = d_points.clone().requires_grad_(True)
    ret, L = ren.forward_with_loss(opt, center, ray, sdf, rad, head, gt, d_points=dp, mask_finish=masks["finish"],
                                   mask_eik=masks["eik"], mask_bg=masks["bg"])
    L["all"].backward()
    cfg = OF.dataset_config("BlendedMVS", dual_field=True, sample_intvs=24, n_levels=8, log2_hashmap_size=14)
    osd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.state_dict().items()}
    ord_ = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in rad.state_dict().items()}
    odp = d_points.cpu().clone().requires_grad_(True)
    oret = OF.render(cfg, center.cpu(), ray.cpu(), osd, ord_)
    lo = loss_head(oret, gt.cpu(), odp, masks["finish"].cpu(), masks["eik"].cpu(), masks["bg"].cpu(), 3.0, 2.0, 0.5)
    lo["all"].backward()
    for k in ("rgb_loss", "eikonal_loss", "DC_loss", "mse", "all"):
        assert abs(float(L[k]) - float(lo[k])) <= 2e-5 * max(1.0, abs(float(lo[k]))), k
    got = _grads(sdf, rad)
    for pre, st in (("s.", osd), ("r.", ord_)):
        for k, v in st.items():
            ref = v.grad if v.grad is not None else torch.zeros_like(v)
            assert rel_err(got[pre + k], ref) < (5e-4 if k == "beta" else 1e-4), pre + k
    assert rel_err(dp.grad.cpu(), odp.grad) < 1e-5


@pytest.mark.parametrize("mode,fused_form", [("allreduce", True), ("allreduce", False), ("uniform", True)])
def test_sharded_loss_head_gives_the_single_process_gradients(mode, fused_form, monkeypatch):
    """two ranks, rays split between them, masks unbalanced.  Each rank's backward must divide by the
    GLOBAL counts, so that the SUM of the ranks' gradients (what the gradient all-reduce forms) equals the gradient of the
    single-process run over all rays.  The process group is emulated: pass 1 records each shard's (sum, count) pairs, pass 2
    feeds their total through ls2fm.dist.globalize_loss_sums exactly where the all-reduce sits."""
    from ls2fm import dist as ldist
    n_rays = 128
    opt, sdf, rad, ren, center, ray, gt, d_points, masks = _setup("ETH3D", True, 32, n_rays, 81)
    if mode == "uniform":
        masks = {k: None for k in masks}
    else:
        masks["finish"][:, : n_rays // 2] = False            # every finished ray sits in the second shard
    head = RenderLossHead(DEV, 3.0, 2.0, 1.0, global_counts=mode)
    dref = d_points if mode != "uniform" else None

    def run(sl):
        kw = {("mask_" + k): (None if m is None else m[:, sl]) for k, m in masks.items()}
        dp = None if dref is None else dref[:, sl]
        if fused_form:
            ret, L = ren.forward_with_loss(opt, center[:, sl].contiguous(), ray[:, sl].contiguous(), sdf, rad, head,
                                           gt[:, sl].contiguous(), d_points=dp, **kw)
        else:
            L = head(ren.forward(opt, center[:, sl].contiguous(), ray[:, sl].contiguous(), sdf, rad), gt[:, sl].contiguous(),
                     d_points=dp, **kw)
        return L

    sdf.zero_grad(); rad.zero_grad()
    L = run(slice(None))
    L["all"].backward()
    full, full_terms = _grads(sdf, rad), {k: float(v) for k, v in L.items()}

    shards = (slice(0, n_rays // 2), slice(n_rays // 2, n_rays))
    recorded = []
    monkeypatch.setattr(ldist, "is_distributed", lambda: True)
    monkeypatch.setattr(ldist, "world_size", lambda: 2)
    monkeypatch.setattr(ldist, "globalize_loss_sums", lambda sums, m: recorded.append(sums.clone()))
    for sl in shards:
        run(sl)
    if mode == "uniform" and fused_form:
        # (ABI 9: the fused forward's reduction multiplies its counts by the world size itself -- ls2fm_loss_spec.count_scale --
        # and nothing runs between forward and backward: no collective, no kernel)
        assert not recorded
    total = recorded[0] + recorded[1] if recorded else None
    if mode == "uniform":
        def fake(sums, m):
            assert m == "uniform"
            sums[1::2] *= 2                    # what dist.globalize_loss_sums does for world size 2, without a process group
    else:
        def fake(sums, m):
            assert m == "allreduce"
            sums.copy_(total)
    monkeypatch.setattr(ldist, "globalize_loss_sums", fake)
    acc, terms = None, []
    for sl in shards:
        sdf.zero_grad(); rad.zero_grad()
        L = run(sl)
        L["all"].backward()
        g = _grads(sdf, rad)
        acc = {k: torch.as_tensor(v).double() for k, v in g.items()} if acc is None else \
            {k: acc[k] + torch.as_tensor(v).double() for k, v in g.items()}
        terms.append({k: float(v) for k, v in L.items()})
    for k in full:
        assert rel_err(acc[k], full[k]) < (2e-4 if k == "s.beta" else 2e-5), k
    for k in ("rgb_loss", "eikonal_loss", "DC_loss", "all"):
        if mode == "allreduce":       # every rank reports the global means
            assert abs(terms[0][k] - full_terms[k]) <= 1e-5 * max(1.0, abs(full_terms[k])), k
            assert terms[0][k] == terms[1][k]
        else:                         # a rank's terms are its share of the global means
            assert abs(terms[0][k] + terms[1][k] - full_terms[k]) <= 1e-5 * max(1.0, abs(full_terms[k])), k


def test_no_async_error_after_a_backward_with_in_launch_hand_offs():
    """ABI 9: the sticky error word of the backward's in-launch hand-offs (weight-gradient reduction rows / finalize tasks riding in
    the scatter_fill launch, csrc/side_jobs.h).  A healthy step at the size that takes that path (>= 32 k samples) leaves it clear,
    and a later backward is not refused (LS2FM_ERR_STARVED)."""
    from ls2fm import _lib
    opt, sdf, rad, ren, center, ray, gt, d_points, masks = _setup("ETH3D", True, 128, 512, 91)
    head = RenderLossHead(DEV, 3.0, 2.0, 0.0)
    for _ in range(2):
        sdf.zero_grad(); rad.zero_grad()
        ret, L = ren.forward_with_loss(opt, center, ray, sdf, rad, head, gt, d_points=d_points)
        L["all"].backward()
    torch.cuda.synchronize()
    assert _lib.async_error() == 0
    assert all(torch.isfinite(p.grad).all() for p in list(sdf.parameters()) + list(rad.parameters()) if p.grad is not None)
