"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- CPU restatement of the loss head the reference's stages apply to the
renderer's outputs.  Each line is the reference's own torch expression:

    rgb_loss      pipelines/Camera.py:535        torch_F.l1_loss(rgb, rgbs_gt)
    eikonal_loss  Initialization.py:257-258      l1_loss(norm(normals, dim=-1), ones)          (BA.py:193-194: normals[mask_bg])
    DC_loss       Camera.py:520-532              smooth_l1_loss(d_points[mask_finish], depth_mlp[mask_finish]) or 0
    mse           Camera.py:533                  mse_loss(rgb[mask_bg], rgbs_gt[mask_bg])       (PSNR = -10 log10)
    all           BA.py:206-218                  sum 10**w * loss

Parity: PINNED -- tests/test_oracle_caller_golden.py composes oracle.fields.render + sphere_tracing + this loss head the way
CameraSet.render / BA.compute_loss / summarize_loss do and checks every term, PSNR, the summed loss and all parameter gradients
against tests/golden/caller_*.npz, recorded from those reference functions themselves (tests/golden/make_golden_caller.py).
"""
import torch
import torch.nn.functional as torch_F


def loss_head(ret, rgbs_gt, d_points=None, mask_finish=None, mask_eik=None, mask_bg=None, w_rgb=3.0, w_eikonal=2.0, w_dc=0.0):
    rgb, normals, depth_mlp = ret["rgb"], ret["normals"], ret["depth_mlp"]
    out = {}
    out["rgb_loss"] = torch_F.l1_loss(rgb, rgbs_gt)
    n = normals if mask_eik is None else normals[mask_eik]
    out["eikonal_loss"] = torch_F.l1_loss(torch.norm(n, dim=-1), torch.ones_like(torch.norm(n, dim=-1)))
    if d_points is None:
        out["DC_loss"] = torch.zeros((), dtype=rgb.dtype)
    else:
        d_points = d_points.view(*depth_mlp.shape)
        m = torch.ones_like(depth_mlp, dtype=torch.bool) if mask_finish is None else mask_finish.view(*depth_mlp.shape)
        if m.sum() > 0:
            out["DC_loss"] = torch_F.smooth_l1_loss(d_points[m], depth_mlp[m], reduction="mean")
        else:
            out["DC_loss"] = torch.zeros_like(d_points).mean()
    if mask_bg is None:
        out["mse"] = torch_F.mse_loss(rgb, rgbs_gt)
    else:
        out["mse"] = torch_F.mse_loss(rgb[mask_bg], rgbs_gt[mask_bg])
    total = 0.0
    for key, w in (("rgb_loss", w_rgb), ("eikonal_loss", w_eikonal), ("DC_loss", w_dc)):
        if w is not None:
            total = total + 10 ** float(w) * out[key]
    out["all"] = total
    return out
