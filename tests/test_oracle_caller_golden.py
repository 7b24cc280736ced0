"""Pin the CPU oracle's CALLER-LEVEL chain -- oracle.fields.render + oracle.fields.sphere_tracing + oracle.losses.loss_head,
composed the way the reference's CameraSet.render / BA.compute_loss / summarize_loss compose them (pipelines/Camera.py:
500-536, BA.py:186-218) -- against tests/golden/caller_*.npz, which were recorded from those reference functions themselves
(tests/golden/make_golden_caller.py).  Pure CPU: this is what makes oracle/losses.py a golden-pinned checker."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_cfg, golden_state, load_golden, rel_err
from oracle import fields as F
from oracle import losses as OL


@pytest.mark.parametrize("case", ["caller_dtu_dual", "caller_eth3d_single"])
def test_oracle_caller_chain_vs_reference(case):
    g = load_golden(case)
    meta = json.loads(bytes(g["meta_json"]).decode())
    meta["bg_sdf"] = None
    cfg = golden_cfg(meta)
    w = meta["weights"]
    sdf_sd = golden_state(g, "sdf", requires_grad=True)
    rad_sd = golden_state(g, "rad", requires_grad=True)
    centers, rays, gt = (torch.from_numpy(g[k]) for k in ("centers", "rays", "rgbs_gt"))
    ret = F.render(cfg, centers, rays, sdf_sd, rad_sd)
    for k in ("rgb", "depth_mlp", "normal_mlp", "sdfs_volume", "normals"):
        assert rel_err(ret[k], g[f"ret/{k}"]) < 2e-6, k
    # Camera.py:506-531: trace every picked ray; mask_bg from the target's grey level; DC over finished, in-mask rays
    d_points, _, _, finish, _ = F.sphere_tracing(cfg, centers.view(1, -1, 3), rays.view(1, -1, 3), sdf_sd, rng=False)
    grey = gt.mean(dim=-1)
    mask_bg = (grey < 0.95) & (grey > 0.05)
    assert np.array_equal(mask_bg.numpy(), g["mask_bg"])
    depth = ret["depth_mlp"]
    mask_finish = finish.view(*depth.shape) & mask_bg.view(*depth.shape)
    out = OL.loss_head(ret, gt, d_points=d_points.view(*depth.shape), mask_finish=mask_finish, mask_eik=mask_bg, mask_bg=mask_bg,
                       w_rgb=w["rgb"], w_eikonal=w["eikonal_loss"], w_dc=w["DC_Loss"])
    psnr = -10 * torch.log10(out["mse"])
    for name, val in (("rgb_loss", out["rgb_loss"]), ("eikonal_loss", out["eikonal_loss"]), ("DC_loss", out["DC_loss"]),
                      ("PSNR", psnr), ("loss_all", out["all"])):
        a, b = float(val.detach()), float(g[name])
        assert abs(a - b) <= 5e-6 * max(abs(b), 1e-3), (name, a, b)
    out["all"].backward()
    for pre, sd in (("sdf", sdf_sd), ("rad", rad_sd)):
        for k, v in sd.items():
            key = f"grad/{pre}/{k}"
            if key not in g:
                continue
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            # beta: an ill-conditioned sum; the oracle repeats the reference's fp32 op order, so it lands close
            tol = 2e-3 if k == "beta" else 5e-5
            assert rel_err(got, g[key]) < tol, (pre, k, rel_err(got, g[key]))


@pytest.mark.parametrize("case", ["points_step_dtu", "points_step_eth3d"])
def test_oracle_points_step_vs_reference(case):
    """the point side of a BA iteration (get_surface_pts -> infer_sdf -> sdf_surf / eikonal, BA.py:117-131, 199-202) as the
    reference's own classes computed it (tests/golden/make_golden_points_step.py)"""
    g = load_golden(case)
    meta = json.loads(bytes(g["meta_json"]).decode())
    meta["bg_sdf"] = None
    cfg = golden_cfg(meta)
    w = meta["weights"]
    table = cfg.table()
    sd = golden_state(g, "sdf", requires_grad=True)
    xyzs = torch.from_numpy(g["xyzs"]).clone().requires_grad_(True)
    xyzs_new, nlen = F.get_surface_pts(xyzs, sd, cfg, table)
    sdfs = F.infer_sdf(xyzs_new, sd, cfg, table, "ret_sdf").view(-1, 1)
    assert rel_err(xyzs_new, g["xyzs_new"]) < 2e-6 and rel_err(nlen, g["normals_value"]) < 2e-6 and rel_err(sdfs, g["sdfs"]) < 2e-5
    assert np.array_equal((sdfs.abs() < 2 * float(g["sdf_threshold"])).numpy(), g["mask_surf"])
    sdf_surf, eik = sdfs.abs().mean(), (nlen - 1).abs().mean()
    total = 10 ** w["sdf_surf"] * sdf_surf + 10 ** w["eikonal_loss"] * eik
    for name, val in (("sdf_surf", sdf_surf), ("eikonal_loss", eik), ("loss_all", total)):
        assert abs(float(val.detach()) - float(g[name])) <= 5e-6 * max(abs(float(g[name])), 1e-3), name
    total.backward()
    assert rel_err(xyzs.grad, g["grad/xyzs"]) < 5e-5
    for k, v in sd.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert rel_err(got, g[f"grad/sdf/{k}"]) < 5e-5, k
