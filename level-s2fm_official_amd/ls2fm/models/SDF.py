"""SDF field with the reference's class surface (models/SDF.py): hash-grid + tiny MLP -> (sdf, 16-d
feature), learnable VolSDF beta, normals, surface projection and bidirectional sphere tracing.

Two execution forms share the same Parameters:
  * general / autograd-composed: HIP hash-grid op (first + second order autograd) + torch dense layers.
    Used whenever a graph is needed (infer_sdf with grad, `gradient`, `get_surface_pts`, the
    differentiable tail of `sphere_tracing`).
  * fused kernels (ls2fm.fused): no-grad `infer_sdf`, the sphere-tracing root-find loop, and -- through
    Renderer.forward -- the whole render forward/backward.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .. import dist as ldist
from .. import fused
from ..util_layers import get_layer_dims
from ..utils.custom_functions import RayAABBIntersector
from .base import Geometry, get_Embedder


class SDF(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        lo = torch.tensor(np.array(opt.data.bound_min), dtype=torch.float32)[None, None, :]
        hi = torch.tensor(np.array(opt.data.bound_max), dtype=torch.float32)[None, None, :]
        # plain tensors in the reference (created on opt.device); buffers here so .to() moves them,
        # non-persistent so the state_dict keeps the reference's keys
        self.register_buffer("bound_max", hi, persistent=False)
        self.register_buffer("bound_min", lo, persistent=False)
        self.register_buffer("center", (hi + lo) / 2, persistent=False)
        self.register_buffer("half_size", (hi - lo) / 2, persistent=False)

        vol = opt.SDF.VolSDF
        self.rescale = vol.rescale
        self.beta_speed = vol.beta_speed
        self.beta = nn.Parameter(torch.tensor([math.log(vol.beta_init) / self.beta_speed], dtype=torch.float32))
        self.sdf_threshold = float(vol.sdf_threshold)
        self.iters_max = int(vol.iters_max_st)
        self.scale_mlp = opt.SDF.NN_Init.scale_mlp
        # grad-enabled infer_sdf / gradient / get_surface_pts: "fused" = one autograd node per call (ls2fm_sdf_eval forward,
        # ls2fm_sdf_points_bwd backward), "composed" = HIP hash-grid op + torch layers + autograd's double backward
        self.point_queries = "fused"
        self.define_network(opt)

    def define_network(self, opt):
        self.embed_fn = get_Embedder(opt=opt, input_dim=3)
        self.SDF_MLP = Geometry(opt=opt, input_dim=self.embed_fn.out_dim, skip=opt.SDF.arch.skip,
                                tf_init=opt.SDF.NN_Init.tf_init, layers=get_layer_dims(opt.SDF.arch.layers))

    # ------------------------------------------------------------------ field evaluation
    def _signed(self, feat, xyz):
        scale = torch.full((), float(self.scale_mlp), device=feat.device)      # true division on every device
        if self.opt.data.inside == True:  # noqa: E712  (opt values may be yaml scalars)
            sdf = feat[..., :1] / scale
            if self.opt.data.bg_sdf == True:  # noqa: E712
                sdf = torch.min(sdf, self.opt.data.bg_rad - xyz.norm(dim=-1, keepdim=True))
            return sdf
        return -feat[..., :1] / scale

    def infer_sdf(self, xyz, mode="ret_sdf"):
        if fused.can_eval_without_graph(self, xyz):
            sdf, feat = fused.sdf_eval(self, xyz, want_feat=(mode != "ret_sdf"))
        elif self.point_queries == "fused" and fused.can_query_points(self, xyz):
            sdf, feat, _ = fused.query_points(self, xyz, want_feat=(mode != "ret_sdf"))      # one node, fused backward
        else:
            enc = self.embed_fn(xyz, rescale=self.rescale, bound_min=self.bound_min, bound_max=self.bound_max)
            feat = self.SDF_MLP(enc)
            sdf = self._signed(feat, xyz)
        if mode == "ret_sdf":
            return sdf
        if mode == "ret_feat":
            return feat
        if mode == "ret_all":
            return sdf, feat
        raise ValueError(f"unknown mode {mode!r}")

    def forward_ab(self):
        beta = torch.exp(self.beta * self.beta_speed)
        return 1.0 / beta, beta

    def sdf_to_sigma(self, sdf, alpha, beta):
        half_lap = 0.5 * torch.exp(-sdf.abs() / beta)
        return alpha * torch.where(sdf >= 0, half_lap, 1 - half_lap)

    def density(self, xyzs):
        alpha, beta = self.forward_ab()
        return self.sdf_to_sigma(self.infer_sdf(xyzs, mode="ret_sdf"), alpha, beta)

    def gradient(self, p):
        """d sdf / d p, itself differentiable (callers put its norm inside losses)."""
        with torch.enable_grad():                       # also under no_grad, as the reference (SDF.py:103)
            p.requires_grad_(True)                      # the reference marks its argument (SDF.py:104, SURVEY C-9)
            if self.point_queries == "fused" and fused.can_query_points(self, p):
                return fused.query_points(self, p, want_normal=True)[2]  # analytic normal; its backward is the fused one too
            y = self.infer_sdf(p, mode="ret_sdf")
            (g,) = torch.autograd.grad(outputs=y, inputs=p, grad_outputs=torch.ones_like(y), create_graph=True,
                                       retain_graph=True, only_inputs=True, allow_unused=True)
        return g

    def get_surface_pts(self, pts):
        if (self.point_queries == "fused" and torch.is_grad_enabled() and not pts.requires_grad and pts.is_cuda
                and pts.dtype == torch.float32 and fused.can_query_points(self, pts) and not fused.can_eval_without_graph(self, pts)):
            # the points carry no graph (BA's tracked points): the value and the normal come from ONE query node -- one forward,
            # one backward pipeline instead of two over the same points (the sum of the two nodes' parameter gradients).
            # The PARAMETER gradients are those of the two-node form; `pts` is marked afterwards, as gradient() marks its argument,
            # but on this path `pts.grad` only receives the direct term of pts - n / |n| * sdf (the points had no graph when the
            # field was queried: the second-order path through the normals is not recorded) -- a caller that needs d / d pts
            # marks the points BEFORE the call and gets the two-node form below.  fp32 only: other dtypes take the general form.
            sdf, _, normals = fused.query_points(self, pts, want_normal=True)
            pts.requires_grad_(True)                    # gradient() marks its argument (SDF.py:104, SURVEY C-9): same side effect
            return fused.surface_points(pts, normals, sdf)
        sdf = self.infer_sdf(pts.detach(), mode="ret_sdf")
        normals = self.gradient(pts)
        wants_graph = pts.requires_grad or normals.requires_grad or sdf.requires_grad
        if self.point_queries == "fused" and wants_graph and pts.is_cuda and pts.dtype == torch.float32 and fused.available(self, pts):
            return fused.surface_points(pts, normals, sdf)          # the two lines below as one node each way
        length = torch.norm(normals, dim=-1, keepdim=True)
        return pts - normals / length.detach() * sdf, length

    # ------------------------------------------------------------------ sphere tracing
    def _trace_loop_torch(self, o, d, near, far):
        """The reference's no-grad root-find loop written with torch ops over the HIP field evaluation
        (kept as the cross-check of the fused kernel; semantics: SURVEY.md Appendix A.5)."""
        thr = self.sdf_threshold
        t_s, t_e = near.clone(), far.clone()
        p_s = o + t_s[:, None] * d
        p_e = o + t_e[:, None] * d
        s_s = self.infer_sdf(p_s)[:, 0].clone()
        s_e = self.infer_sdf(p_e)[:, 0].clone()
        live_s = live_e = None
        track, t_end = [], [t_e]
        trips = 0
        while True:
            s_s = torch.where(s_s.abs() <= thr, torch.zeros_like(s_s), s_s)
            s_e = torch.where(s_e.abs() <= thr, torch.zeros_like(s_e), s_e)
            live_s = (s_s.abs() > thr) if live_s is None else live_s & (s_s.abs() > thr)
            live_e = (s_e.abs() > thr) if live_e is None else live_e & (s_e.abs() > thr)
            # the reference's break tests a mask over ALL rays (SDF.py:167): sharded runs reduce it over the ranks, one
            # collective per trip -- every rank must take this same (torch) path, as with the fused kernel's single max-reduce
            if trips == self.iters_max or not ldist.global_any(live_s.any()):
                break
            trips += 1
            t_s = t_s + s_s
            t_e = t_e + s_e
            t_s = torch.where(t_s > far, far, t_s)
            t_e = torch.where(t_e > far, far, t_e)
            track.append(p_s)
            p_s = o + t_s[:, None] * d
            p_e = o + t_e[:, None] * d
            if bool(live_s.any()):
                s_s = s_s.clone()
                s_s[live_s] = self.infer_sdf(p_s[live_s])[:, 0]
            if bool(live_e.any()):
                s_e = s_e.clone()
                s_e[live_e] = self.infer_sdf(p_e[live_e])[:, 0]
            ordered = t_s < t_e
            live_s, live_e = live_s & ordered, live_e & ordered
            t_end.append(t_e)
        if not track:
            track = [p_s]
        return torch.stack(track, dim=1), t_end[-1], trips

    def _sphere_tracing_static(self, o, d, shape2, rgbs_gt=None, want_samples=False, launch_stream=None, sample_u=None):
        """sphere_tracing without the host round trip for the trip count K (hipGraph-capturable, ls2fm.stage): the kernel leaves K
        on the device and runs every ray for iters_max trips anyway; the differentiable depth sums the first K track points
        through a device-side mask (K = 0: the single current point, SDF.py:201-202) -- one fused node, ls2fm.fused.traced_depth.
        Same d_pred / sdf_last / finish_mask as the synchronising form.  The RNG-dependent `sampled_pts` (SDF.py:216-224) has
        a K-dependent SHAPE in the reference; with want_samples it comes back at its largest shape, [1, min(4096, R) * iters_max
        + R, 3] -- the track points of up to 4096 random rays, then one random point per ray between near and 1.5 x the far-end
        distance -- with `self.last_sample_mask` ([same] bool, device) marking the rows the reference would have returned (the
        first K columns of every picked track); the draws come from the device generator, so the call stays capturable.
        Otherwise None.  rgbs_gt [.., 3]: the node also forms CameraSet.render's mask_bg and mask_finish & mask_bg
        (Camera.py:515-516) -> self.last_masks."""
        if not fused.available(self, o):
            raise RuntimeError("ls2fm: static_trips needs the fused tracing kernel (GPU tensors, reference layer sizes)")
        # launch_stream: the kernels of this call (root-find, track evaluation, depth) are enqueued on that stream -- which the
        # caller has made wait for the inputs and whose completion it waits for itself -- while the autograd node stays with the
        # current stream (ls2fm.stage: tracing beside the render's gather pass)
        with torch.no_grad():
            near, far, track, t_end, trips = fused.sphere_trace(self, o.detach(), d.detach(), sync=False, launch_stream=launch_stream)
        d_pred, last, finish, mask_bg, mask_dc = fused.traced_depth(self, track, trips, near, far, rgbs_gt,
                                                                    trace_ws=getattr(track, "_ls2fm_trace_ws", None),
                                                                    launch_stream=launch_stream)
        self.last_trips = trips
        self.last_masks = (mask_bg, mask_dc)             # uint8 [R] each (rgbs_gt given): what the fused loss head takes
        sampled = None
        self.last_sample_mask = None
        if want_samples:
            # t_end / trips / track / near / far are written by kernels on `launch_stream` when one is given: the sample block is
            # enqueued there too (the caller joins that stream before it reads any output of this call)
            import contextlib
            on_stream = torch.cuda.stream(launch_stream) if launch_stream is not None else contextlib.nullcontext()
            with torch.no_grad(), on_stream:
                n_rays, it = o.shape[0], int(self.iters_max)
                k = trips.long().reshape(1)
                # sample_u [R]: the draw of SDF.py:217 given by the caller (parity tests replay the reference's)
                u = torch.rand(n_rays, device=o.device) if sample_u is None else sample_u.reshape(-1).to(o.device)
                t_up = torch.minimum(1.5 * t_end.gather(1, k.expand(n_rays, 1))[:, 0], far)          # far end after K trips
                along = o + ((1 - u) * t_up + u * near)[:, None] * d
                pick = torch.randperm(n_rays, device=o.device)[:4096]
                cols = torch.arange(it, device=o.device)[None, :] < k.clamp_min(1)                    # K = 0: the current point
                sampled = torch.cat([track[pick, :it].reshape(1, -1, 3), along.view(1, -1, 3)], dim=1)
                self.last_sample_mask = torch.cat([cols.expand(pick.shape[0], it).reshape(-1),
                                                   torch.ones(n_rays, dtype=torch.bool, device=o.device)])
        return d_pred.view(*shape2), last, sampled, finish.view(-1, 1)

    def sphere_tracing(self, ray0, ray_direction, model=None, c=None, tau=0.5, n_steps=(128, 129),
                       n_secant_steps=8, depth_range=(0.0, 2.4), max_points=3500000, rad=1.0, iter=0,
                       impl="fused", static_trips=False, rgbs_gt=None, want_samples=False, launch_stream=None, sample_u=None):
        """ray0, ray_direction [B,R,3] -> (d_pred [B,R], sdf_last [B*R], sampled_pts [1, <=4096+B*R, 3],
        finish_mask [B*R,1]).  `d_pred = near + sum_k sdf(track_k)` is differentiable w.r.t. the SDF
        parameters; the root-find itself runs without a graph.  Unused reference arguments are accepted."""
        shape2 = ray_direction.shape[:2]
        o = ray0.reshape(-1, 3)
        d = ray_direction.reshape(-1, 3)
        if static_trips:
            return self._sphere_tracing_static(o, d, shape2, rgbs_gt, want_samples=want_samples, launch_stream=launch_stream,
                                               sample_u=sample_u)
        with torch.no_grad():
            if impl == "fused" and fused.available(self, o):
                near, far, pts_tracks, t_end, trips = fused.sphere_trace(self, o.detach(), d.detach())
            else:
                _, hits_t, _ = RayAABBIntersector.apply(o, d, self.center.view(1, 3), self.half_size.view(1, 3), 1)
                near, far = hits_t[:, 0, 0], hits_t[:, 0, 1]
                pts_tracks, t_end, trips = self._trace_loop_torch(o.detach(), d.detach(), near, far)
        self.last_trips = trips
        sdf_tracks = self.infer_sdf(pts_tracks.detach(), mode="ret_sdf")          # graph-enabled  [R,K,1]
        d_pred = sdf_tracks.sum(dim=-2).view(*shape2) + near.view(*shape2)
        far2 = far.view(*shape2)
        d_pred = torch.where(d_pred > far2, far2, d_pred)
        extent = self.bound_max.reshape(-1)[0] - self.bound_min.reshape(-1)[0]
        finish_mask = sdf_tracks[:, -1, :].abs() < extent / 10 / self.opt.Res
        # random eikonal sample points (RNG stays on the host side in torch, same call order as the reference)
        u = torch.rand_like(d_pred) if sample_u is None else sample_u.reshape(d_pred.shape).to(d_pred.device)
        t_up = 1.5 * t_end.view(*shape2)
        t_up = torch.where(t_up > far2, far2, t_up)
        t_rand = (1 - u) * t_up + u * near.view(*shape2)
        sampled = ray0 + t_rand[..., None] * ray_direction
        pick = torch.randperm(pts_tracks.shape[0])[:4096]
        sampled = torch.cat([pts_tracks[pick.to(pts_tracks.device)].view(1, -1, 3), sampled.view(1, -1, 3)], dim=1)
        return d_pred, sdf_tracks[:, -1, 0], sampled, finish_mask
