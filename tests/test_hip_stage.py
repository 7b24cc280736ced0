"""The stage drivers' optimisation step (ls2fm.stage) against
  (a) CALLER-LEVEL GOLDENS recorded from the reference's own CameraSet.render + BA.compute_loss / summarize_loss + backward
      (tests/golden/make_golden_caller.py: pipelines/Camera.py:448-538, BA.py:186-218) with a fixed ray pick,
  (b) a plain-torch restatement of the same step (composed render, torch losses, torch.optim.Adam + ExponentialLR) over a
      short trajectory -- eagerly and as ONE captured hipGraph per step (tracing, point queries, fused render + loss head,
      backward, Adam with the schedule on the device)."""
import json
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from helpers import named_grads, options_for
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from ls2fm import stage
from ls2fm.losses import RenderLossHead
from test_hip_fused_render import _beta_ok

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _product(g):
    meta = json.loads(bytes(g["meta_json"]).decode())
    meta["bg_sdf"] = None
    opt = options_for(meta, DEV)
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    sdf.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sdf/")}, strict=True)
    rad.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("rad/")}, strict=True)
    return meta, opt, sdf, rad, ren


def _exact_beta(meta, sdf, rad, centers, rays, gt, ret, w):
    """d loss_all / d beta of the fp32 computation with everything beta enters carried out in fp64
    (oracle.fields.beta_gradient_exact_sum); the traced depth and the masks are those of the run under test"""
    from conftest import golden_cfg
    from oracle import fields as OF
    cfg = golden_cfg(dict(meta, bg_sdf=None))
    gt64 = gt.detach().cpu().double()
    m_fin = ret["mask_finish"].detach().cpu()
    d_pts = ret["d_points"].detach().cpu().double()

    def loss_fn(out):
        total = 10 ** w["rgb"] * (out["rgb"] - gt64).abs().mean()
        if m_fin.any():
            total = total + 10 ** w["DC_Loss"] * torch.nn.functional.smooth_l1_loss(d_pts[m_fin], out["depth_mlp"][m_fin])
        return total
    sd = {k: v.detach().cpu().float() for k, v in sdf.state_dict().items()}
    rd = {k: v.detach().cpu().float() for k, v in rad.state_dict().items()}
    return OF.beta_gradient_exact_sum(cfg, centers.detach().cpu(), rays.detach().cpu(), sd, rd, loss_fn, with_condition=True)


@pytest.mark.parametrize("case", ["caller_dtu_dual", "caller_eth3d_single"])
@pytest.mark.parametrize("static_trips", [False, True])
def test_render_losses_vs_reference_caller_goldens(case, static_trips):
    g = load_golden(case)
    meta, opt, sdf, rad, ren = _product(g)
    w = meta["weights"]
    head = RenderLossHead(DEV, w_rgb=w["rgb"], w_eikonal=w["eikonal_loss"], w_dc=w["DC_Loss"])
    centers, rays = torch.from_numpy(g["centers"]).to(DEV), torch.from_numpy(g["rays"]).to(DEV)
    gt = torch.from_numpy(g["rgbs_gt"]).to(DEV)
    ret = stage.render_losses(opt, ren, sdf, rad, head, centers, rays, gt, static_trips=static_trips)
    for k in ("rgb", "depth_mlp", "normal_mlp", "sdfs_volume", "normals"):
        assert rel_err(ret[k].cpu(), g[f"ret/{k}"]) < 2e-5, k
    assert np.array_equal(ret["mask_bg"].cpu().numpy(), g["mask_bg"])
    for k, key in (("rgb_loss", "rgb_loss"), ("DC_loss", "DC_loss"), ("PSNR", "PSNR"), ("eikonal_loss", "eikonal_loss"),
                   ("loss_all", "loss_all")):
        a, b = float(ret[k]), float(g[key])
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
    ret["loss_all"].backward()
    for pre, mod in (("sdf", sdf), ("rad", rad)):
        for k, v in named_grads(mod).items():
            if k == "beta":             # one ill-conditioned scalar: adjudicated against its exactly summed value
                exact, cond = _exact_beta(meta, sdf, rad, centers, rays, gt, ret, w)
                print(f"[{case}] d beta: condition number of the sum {cond:.3g}")
                _beta_ok(v, g["grad/sdf/beta"], exact, condition=cond)
                continue
            assert rel_err(v, g[f"grad/{pre}/{k}"]) < 1e-4, (pre, k)


def _torch_step(opt, ren, sdf, rad, optim, sched, centers, rays, gt, w):
    """the reference's lines, composed form + torch ops"""
    optim.zero_grad(set_to_none=True)
    ret = ren.forward_composed(opt, centers, rays, sdf, rad)
    d_points, _, _, mask_finish = sdf.sphere_tracing(centers.view(1, -1, 3), rays.view(1, -1, 3), sdf, impl="torch")
    depth = ret["depth_mlp"]
    d_points = d_points.view(*depth.shape)
    gray = gt.mean(dim=-1)
    mask_bg = (gray < 0.95) & (gray > 0.05)
    mask_finish = mask_finish.view(*depth.shape) & mask_bg.view(*depth.shape)
    if mask_finish.sum() > 0:
        dc = torch.nn.functional.smooth_l1_loss(d_points[mask_finish], depth[mask_finish], reduction="mean")
    else:
        dc = torch.zeros_like(d_points).mean()
    rgb_loss = torch.nn.functional.l1_loss(ret["rgb"], gt)
    nrm = torch.norm(ret["normals"][mask_bg], dim=-1)
    eik = torch.nn.functional.l1_loss(nrm, torch.ones_like(nrm))
    total = 10 ** w["rgb"] * rgb_loss + 10 ** w["eikonal_loss"] * eik + 10 ** w["DC_Loss"] * dc
    total.backward()
    optim.step()
    sched.step()
    return float(total.detach())


@pytest.mark.parametrize("capture", [False, True])
def test_stage_trajectory_matches_plain_torch(capture):
    g = load_golden("caller_dtu_dual")
    meta, opt, sdf_a, rad_a, ren = _product(g)
    _, _, sdf_b, rad_b, _ = _product(g)
    w = dict(rgb=3, eikonal_loss=1, DC_Loss=0)
    lr, lr_end, iters = 5e-3, 5e-4, 20
    st = stage.RenderStage(opt, ren, sdf_a, rad_a, weights=w, lr=lr, lr_end=lr_end, max_iter=iters, eps=1e-15, capture=capture)
    pb = list(sdf_b.parameters()) + list(rad_b.parameters())
    ob = torch.optim.Adam(pb, lr=lr, eps=1e-15)
    sb = torch.optim.lr_scheduler.ExponentialLR(ob, (lr_end / lr) ** (1.0 / iters))
    gen = torch.Generator().manual_seed(5)
    base_c, base_r = torch.from_numpy(g["centers"]).to(DEV), torch.from_numpy(g["rays"]).to(DEV)
    gt = torch.from_numpy(g["rgbs_gt"]).to(DEV)
    la, lb = [], []
    for it in range(8):
        jitter = (0.01 * torch.randn(base_r.shape, generator=gen)).to(DEV)              # new rays every step
        c, r = base_c, (base_r + jitter).contiguous()
        la.append(float(st.step(c, r, gt)["loss_all"]))
        lb.append(_torch_step(opt, ren, sdf_b, rad_b, ob, sb, c, r, gt, w))
    for x, y in zip(la, lb):
        assert abs(x - y) <= 3e-3 * abs(y), (la, lb)                     # same trajectory (Adam amplifies last-bit noise)
    assert la[-1] < la[0]
    assert abs(st.optim.param_groups[0]["lr"] - ob.param_groups[0]["lr"]) < 1e-12
    assert int(st.optim.state[st.params[0]]["step"]) == 8
    for (n, p), q in zip(list(sdf_a.named_parameters()) + list(rad_a.named_parameters()), pb):
        if not n.endswith("embedder_obj.params"):
            assert rel_err(p.detach().cpu(), q.detach().cpu()) < 5e-2, n


@pytest.mark.parametrize("case", ["points_step_dtu", "points_step_eth3d"])
def test_surface_losses_vs_reference_points_step_goldens(case):
    """stage.surface_losses (fused point queries) against the reference's own get_surface_pts -> infer_sdf -> BA.compute_loss
    ("sfm" branch) -> summarize_loss -> backward (tests/golden/make_golden_points_step.py)"""
    g = load_golden(case)
    meta = json.loads(bytes(g["meta_json"]).decode())
    meta["bg_sdf"] = None
    opt = options_for(meta, DEV)
    opt.Res = meta["Res"]
    sdf = SDF(opt).to(DEV)
    sdf.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sdf/")}, strict=True)
    assert sdf.point_queries == "fused"
    xyzs = torch.from_numpy(g["xyzs"]).to(DEV).requires_grad_(True)
    out = stage.surface_losses(opt, sdf, xyzs)
    assert rel_err(out["xyzs_new"].cpu(), g["xyzs_new"]) < 2e-5
    assert rel_err(out["gradients"].cpu(), g["normals_value"]) < 2e-5
    assert rel_err(out["sdfs"].cpu(), g["sdfs"]) < 5e-5
    differ = (out["mask_surf"].cpu().numpy() != g["mask_surf"])
    near_edge = np.abs(np.abs(g["sdfs"]) - 2 * float(g["sdf_threshold"])) < 1e-5 * np.abs(g["sdfs"]).max()
    assert not (differ & ~near_edge).any()               # the mask is a threshold on sdfs: only ties may flip
    w = meta["weights"]
    total = 10 ** w["sdf_surf"] * out["sdf_surf"] + 10 ** w["eikonal_loss"] * out["eikonal_loss"]
    for name, val in (("sdf_surf", out["sdf_surf"]), ("eikonal_loss", out["eikonal_loss"]), ("loss_all", total)):
        assert abs(float(val.detach()) - float(g[name])) <= 2e-5 * max(abs(float(g[name])), 1e-3), name
    total.backward()
    # d loss / d xyzs is the one output of this step that is ill-conditioned IN ITS INPUT: it contains the field's second
    # derivative (softplus with beta = 100, piecewise-constant grid Hessian) -- moving the input positions by ONE ulp moves the
    # reference's own fp32 result by more than 1e-4 of its scale on a third to a half of the points.  Each point is therefore
    # held to 1e-4 of the scale plus a small multiple of its own one-ulp sensitivity, measured with the CPU oracle.
    # (the parameter gradients inherit it -- a table entry collects one or two of these points -- and get the same treatment)
    sens = _ulp_sensitivity(meta, g, w)
    got = {"xyzs": xyzs.grad.cpu().numpy(), **{"sdf/" + k: v.numpy() for k, v in named_grads(sdf).items()}}
    for k, v in got.items():
        ref = g["grad/" + k]
        err, scale = np.abs(v - ref), np.abs(ref).max()
        assert (err <= 1e-4 * scale + 4.0 * sens[k]).all(), (k, err.max() / scale, float((err > 1e-4 * scale).mean()))
        assert err.max() <= 2e-3 * scale, k


def _ulp_sensitivity(meta, g, w):
    """{name: array}: how far the oracle's fp32 gradients of the point step (d loss / d xyzs and every parameter's) move, element
    by element, when every input position moves by one ulp, either way"""
    from conftest import golden_cfg, golden_state
    from oracle import fields as OF
    cfg = golden_cfg(meta)
    table = cfg.table()

    def grads_at(x0):
        sd = golden_state(g, "sdf", requires_grad=True)
        x = x0.clone().requires_grad_(True)
        xn, nlen = OF.get_surface_pts(x, sd, cfg, table)
        sdfs = OF.infer_sdf(xn, sd, cfg, table, "ret_sdf").view(-1, 1)
        (10 ** w["sdf_surf"] * sdfs.abs().mean() + 10 ** w["eikonal_loss"] * (nlen - 1).abs().mean()).backward()
        out = {"xyzs": x.grad.numpy()}
        for k, v in sd.items():
            out["sdf/" + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
        return out
    x0 = torch.from_numpy(g["xyzs"])
    base = grads_at(x0)
    sens = {k: np.zeros_like(v) for k, v in base.items()}
    for direction in (float("inf"), float("-inf")):
        other = grads_at(torch.nextafter(x0, torch.full_like(x0, direction)))
        for k in sens:
            sens[k] = np.maximum(sens[k], np.abs(other[k] - base[k]))
    return sens


@pytest.mark.parametrize("capture", [False, True])
def test_stage_step_with_point_side_terms(capture):
    """a BA-shaped step: render terms + the point side (surface_losses on tracked points) as `extra_loss`, against the same step
    written with torch ops (composed render, torch losses, composed point queries) + torch.optim.Adam"""
    g = load_golden("caller_dtu_dual")
    meta, opt, sdf_a, rad_a, ren = _product(g)
    _, _, sdf_b, rad_b, _ = _product(g)
    opt.Res = 128
    w = dict(rgb=3, eikonal_loss=1, DC_Loss=0)
    gen = torch.Generator().manual_seed(11)
    xyzs = ((torch.rand(200, 3, generator=gen) * 2 - 1) * 0.7).to(DEV)

    def extra_a(ret):
        out = stage.surface_losses(opt, sdf_a, xyzs)
        return 10.0 * out["sdf_surf"] + 3.0 * out["eikonal_loss"]
    st = stage.RenderStage(opt, ren, sdf_a, rad_a, weights=w, lr=2e-3, lr_end=2e-4, max_iter=10, eps=1e-15, capture=capture,
                           extra_loss=extra_a)
    pb = list(sdf_b.parameters()) + list(rad_b.parameters())
    ob = torch.optim.Adam(pb, lr=2e-3, eps=1e-15)
    sb = torch.optim.lr_scheduler.ExponentialLR(ob, (2e-4 / 2e-3) ** (1.0 / 10))
    sdf_b.point_queries = "composed"
    c, r = torch.from_numpy(g["centers"]).to(DEV), torch.from_numpy(g["rays"]).to(DEV)
    gt = torch.from_numpy(g["rgbs_gt"]).to(DEV)
    la, lb = [], []
    for it in range(5):
        la.append(float(st.step(c, r, gt)["loss_all"]))
        ob.zero_grad(set_to_none=True)
        ret = ren.forward_composed(opt, c, r, sdf_b, rad_b)
        d_points, _, _, mask_finish = sdf_b.sphere_tracing(c.view(1, -1, 3), r.view(1, -1, 3), sdf_b, impl="torch")
        depth = ret["depth_mlp"]
        gray = gt.mean(dim=-1)
        mask_bg = (gray < 0.95) & (gray > 0.05)
        mfin = mask_finish.view(*depth.shape) & mask_bg.view(*depth.shape)
        dc = torch.nn.functional.smooth_l1_loss(d_points.view(*depth.shape)[mfin], depth[mfin]) if mfin.sum() > 0 else depth.sum() * 0
        nrm = torch.norm(ret["normals"][mask_bg], dim=-1)
        total = 10 ** w["rgb"] * torch.nn.functional.l1_loss(ret["rgb"], gt) + 10 ** w["eikonal_loss"] * (nrm - 1).abs().mean() \
            + 10 ** w["DC_Loss"] * dc
        xs, nl = sdf_b.get_surface_pts(xyzs)
        total = total + 10.0 * sdf_b.infer_sdf(xs).abs().mean() + 3.0 * (nl - 1).abs().mean()
        total.backward()
        ob.step(); sb.step()
        lb.append(float(total.detach()))
    for x, y in zip(la, lb):
        assert abs(x - y) <= 3e-3 * abs(y), (la, lb)
    assert "loss_extra" in st.step(c, r, gt)
