"""Volumetric renderer with the reference's class surface (models/Renderer.py).

`forward` = ray/AABB near-far -> N uniform mid-point samples -> SDF (+ analytic normal) -> radiance ->
VolSDF density -> front-to-back alpha composite -> background / expected depth / expected normal.
On MI355X with the reference's network sizes it runs as the fused HIP forward/backward
(ls2fm.fused.render); otherwise as the general autograd composition below (HIP hash-grid op + torch).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import fused
from ..utils import camera
from ..utils.custom_functions import RayAABBIntersector


class Renderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        dev = opt.device
        self.bound_max = torch.tensor(np.array(opt.data.bound_max), dtype=torch.float32, device=dev)[None, None, :]
        self.bound_min = torch.tensor(np.array(opt.data.bound_min), dtype=torch.float32, device=dev)[None, None, :]
        self.center = (self.bound_max + self.bound_min) / 2
        self.half_size = (self.bound_max - self.bound_min) / 2
        try:                                   # per-scene override (e.g. DTU scan37: options/DTU.yaml:21)
            scene_cfg = opt.data[f"{opt.data.scene}"]
        except (KeyError, TypeError):
            scene_cfg = None
        bg = getattr(scene_cfg, "bgcolor", None) if scene_cfg is not None else None
        if bg is None:
            bg = opt.data.bgcolor
        self.bgcolor = torch.tensor(np.array(bg), dtype=torch.float32, device=dev)
        self.bg_host = [float(v) for v in np.array(bg, dtype=np.float64).reshape(-1)]

    # ------------------------------------------------------------------ pieces (same math as the fused kernels)
    def composite(self, ray, rgb_samples, density_samples, depth_samples):
        """ray [B,R,3], rgb [B,R,N,3], density [B,R,N], depth [B,R,N,1] -> rgb [B,R,3], prob [B,R,N-1,1]
        (N-1 intervals: the last sample only feeds the background terms, SURVEY C-8)."""
        seg = (depth_samples[..., 1:, 0] - depth_samples[..., :-1, 0]) * ray.norm(dim=-1, keepdim=True)
        tau = density_samples[..., :-1] * seg
        alpha = 1 - torch.exp(-tau)
        before = torch.cumsum(tau, dim=2) - tau                 # exclusive prefix: optical depth in front
        prob = (torch.exp(-before) * alpha)[..., None]
        return (rgb_samples[..., :-1, :] * prob).sum(dim=2), prob

    def sample_depth(self, opt, min_d=None, max_d=None):
        n = opt.SDF.VolSDF.sample_intvs
        mid = (torch.arange(n, device=min_d.device, dtype=torch.float32) + 0.5)[None, None, :, None]
        near, far = min_d[..., None, :], max_d[..., None, :]
        # tensor / tensor: a true division.  (tensor / python-scalar is lowered to a multiply by the
        # rounded reciprocal on the GPU, 1 ulp away from the CPU result; the normals amplify that.)
        return mid / torch.full((), float(n), device=min_d.device) * (far - near) + near       # [B,R,N,1]

    def sdf_to_sigma(self, sdf, alpha, beta):
        half_lap = 0.5 * torch.exp(-sdf.abs() / beta)
        return alpha * torch.where(sdf >= 0, half_lap, 1 - half_lap)

    def volsdf_sampling(self, opt, center, ray, SDF_Field=None):
        _, hits_t, _ = RayAABBIntersector.apply(center.reshape(-1, 3), ray.reshape(-1, 3), self.center.squeeze(0),
                                                self.half_size.squeeze(0), 1)
        near_far = hits_t.squeeze(-2).view(*center.shape[:2], 2)
        if opt.SDF.VolSDF.volsdf_sampling == False:  # noqa: E712
            t = self.sample_depth(opt, min_d=near_far[..., :1], max_d=near_far[..., 1:]).squeeze(-1)
            return t, t, t
        raise NotImplementedError(
            "VolSDF error-bound up-sampling is off in every shipped config (options/LevelS2fM.yaml:26) and the "
            "reference's branch cannot run as written (SURVEY.md C-5); only uniform sampling is on the hot path")

    # ------------------------------------------------------------------ forward
    def forward(self, opt, center, ray, SDF_Field, Rad_Field):
        plan = fused.render_plan(self, opt, center, ray, SDF_Field, Rad_Field)
        if plan is not None:
            return fused.render(self, opt, center, ray, SDF_Field, Rad_Field, plan=plan)
        return self.forward_composed(opt, center, ray, SDF_Field, Rad_Field)

    def forward_with_loss(self, opt, center, ray, SDF_Field, Rad_Field, head, rgbs_gt, d_points=None, mask_finish=None,
                          mask_eik=None, mask_bg=None, inputs_ready=None, depth_node=None):
        """`Renderer.forward` followed by the stage loops' loss head (ls2fm.losses.RenderLossHead: pipelines/Camera.py:
        515-537, BA.py:206-218) -> (ret, losses).  On the fused path the loss head runs INSIDE the render: partial sums in
        the forward's last kernel, the upstream of rgb / normals / depth formed in the backward's first -- no loss kernels and
        no [B,R,N,3] gradient tensor between forward and backward.  Same values and gradients as `head(self.forward(...),
        rgbs_gt, ...)`, which is what runs when the configuration is served by the composed form.
        mask_eik / mask_bg: per-ray masks, None (every ray) or "gt" = CameraSet.render's mask_bg, formed from rgbs_gt (inside the
        fused kernels).  inputs_ready: a torch.cuda.Event recorded (on another stream) once d_points and mask_finish are final --
        the fused forward waits for it only in front of its loss reduction, behind the gather pass and the shading
        (ls2fm_render_opts.loss_inputs_ready; mask_eik / mask_bg tensors must then be final already), the composed form right away."""
        plan = fused.render_plan(self, opt, center, ray, SDF_Field, Rad_Field)
        if plan is not None:
            spec = head.spec(rgbs_gt, mask_finish=mask_finish, mask_eik=mask_eik, mask_bg=mask_bg,
                             n_rays=center.shape[0] * center.shape[1])
            spec.ready = inputs_ready
            spec.depth_node = depth_node        # ls2fm.fused.TracedDepthNode: its backward runs beside this render's scatter
            ret = fused.render(self, opt, center, ray, SDF_Field, Rad_Field, loss=spec, d_points=d_points, plan=plan)
            losses = head.as_dict(ret.pop("loss_terms"), ret.pop("loss_total"))
            losses["PSNR"] = ret.pop("loss_psnr")          # -10 log10(mse), formed with the terms (no graph)
            return ret, losses
        if inputs_ready is not None:
            torch.cuda.current_stream(center.device).wait_event(inputs_ready)
        ret = self.forward_composed(opt, center, ray, SDF_Field, Rad_Field)
        return ret, head(ret, rgbs_gt, d_points=d_points, mask_finish=mask_finish, mask_eik=mask_eik, mask_bg=mask_bg)

    def forward_composed(self, opt, center, ray, SDF_Field, Rad_Field):
        """General autograd composition (pose gradients, arbitrary layer sizes, second-order use)."""
        t, _, _ = self.volsdf_sampling(opt, center, ray, SDF_Field=SDF_Field)
        t = t[..., None]                                                            # [B,R,N,1]
        pts = camera.get_3D_points_from_depth(opt, center, ray, t, multi_samples=True)
        alpha, beta = SDF_Field.forward_ab()
        sdfs, feats = SDF_Field.infer_sdf(pts, mode="ret_all")
        normals = SDF_Field.gradient(pts)
        view = Rad_Field.infer_embed_v(ray_utils=ray[..., None, :].expand_as(pts))
        geo = feats[..., 1:]
        if opt.Ablate_config.dual_field == True:  # noqa: E712
            geo = torch.cat([geo, Rad_Field.Geometry_feat(pts)[..., 1:]], dim=-1)
        rgbs = Rad_Field.infer_app(torch.cat([pts, normals, view, geo], dim=-1))
        sigma = SDF_Field.sdf_to_sigma(sdf=sdfs, alpha=alpha, beta=beta)
        rgb, prob = self.composite(ray=ray, rgb_samples=rgbs, density_samples=sigma.squeeze(-1), depth_samples=t)
        return self._epilogue(rgb, prob, t, sdfs, normals)

    def _epilogue(self, rgb, prob, t, sdfs, normals):
        opacity = prob.sum(dim=2)                                                   # [B,R,1]
        leftover = 1 - opacity
        bg = self.bgcolor.reshape(*([1] * (rgb.dim() - 1)), 3).to(rgb.device)
        self.bgcolor = bg                     # the reference leaves bgcolor reshaped after a call (SURVEY C-3)
        return {
            "rgb": rgb + leftover * bg,
            "sdfs_volume": sdfs,
            "normals": normals,
            "depth_mlp": (t[..., :-1, :] * prob).sum(dim=2) + leftover * t[..., -1, :],
            "normal_mlp": (normals[..., :-1, :] * prob).sum(dim=2) + leftover * normals[..., -1, :],
        }

    # ------------------------------------------------------------------ VolSDF sampler helpers (off the live path)
    def error_bound(self, d_vals, sdf, alpha, beta):
        """VolSDF opacity-approximation error bound per interval: d_vals, sdf [..., M] -> [..., M-1]."""
        sigma = self.sdf_to_sigma(sdf, alpha, beta)
        delta = d_vals[..., 1:] - d_vals[..., :-1]
        tau = sigma[..., :-1] * delta
        depth_before = torch.cumsum(tau, dim=-1) - tau
        mag = sdf.abs()
        d_star = torch.clamp_min(0.5 * (mag[..., :-1] + mag[..., 1:] - delta), 0.0)
        err = alpha / (4 * beta) * delta ** 2 * torch.exp(-d_star / beta)
        bound = torch.exp(-depth_before) * (torch.exp(torch.cumsum(err, dim=-1)) - 1.0)
        return torch.where(torch.isnan(bound), torch.full_like(bound, float("inf")), bound)

    def sample_pdf(self, bins, weights, N_importance, det=False, eps=1e-5):
        """inverse-CDF sampling of `bins` [..., M] with per-interval `weights` [..., M-1]."""
        w = weights + 1e-5
        cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
        cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
        if det:
            u = torch.linspace(0.0, 1.0, N_importance, device=w.device).expand(*cdf.shape[:-1], N_importance)
        else:
            u = torch.rand(*cdf.shape[:-1], N_importance, device=w.device)
        u = u.contiguous()
        hi = torch.searchsorted(cdf.detach(), u, right=False)
        lo = (hi - 1).clamp_min(0)
        hi = hi.clamp_max(cdf.shape[-1] - 1)
        c0, c1 = cdf.gather(-1, lo), cdf.gather(-1, hi)
        b0, b1 = bins.gather(-1, lo), bins.gather(-1, hi)
        span = c1 - c0
        span = torch.where(span < eps, torch.ones_like(span), span)
        return b0 + (u - c0) / span * (b1 - b0)

    def sample_depth_from_opacity(self, opt, depth_sample, opacity_approx):
        n_final = opt.SDF.VolSDF.final_sample_intvs
        cdf = torch.cat([torch.zeros_like(opacity_approx[..., :1]), opacity_approx], -1)
        edges = torch.linspace(0, 1, n_final + 1, device=cdf.device)
        u = (0.5 * (edges[:-1] + edges[1:])).expand(*cdf.shape[:-1], n_final).contiguous()
        hi = torch.searchsorted(cdf, u, right=False)
        lo = (hi - 1).clamp_min(0)
        hi = hi.clamp_max(cdf.shape[-1] - 1)
        d0, d1 = depth_sample.gather(-1, lo), depth_sample.gather(-1, hi)
        c0, c1 = cdf.gather(-1, lo), cdf.gather(-1, hi)
        return (d0 + (u - c0) / (c1 - c0 + 1e-8) * (d1 - d0))[..., None]

    def opacity_to_sample(self, opt, depth_samp, sdf, alpha, beta, B=None, HW=None):
        sigma = self.sdf_to_sigma(sdf, alpha, beta)
        tau = sigma[..., :-1] * (depth_samp[..., 1:] - depth_samp[..., :-1])
        before = torch.cat([torch.zeros_like(tau[..., :1]), torch.cumsum(tau, dim=-1)], dim=-1)[..., :-1]
        return self.sample_depth_from_opacity(opt, depth_samp, 1 - torch.exp(-before))
