"""The optimisation step the reference's stage drivers share:

    Initializer.run        pipelines/Initialization.py:149-179      BA.run_ba (mode != "sfm")   pipelines/BA.py:117-182
    Refine.run             pipelines/rendering_refine.py:78-96

each iteration of which is  `CameraSet.render` (pipelines/Camera.py:448-538: Renderer.forward, SDF.sphere_tracing, mask_bg /
mask_finish, rgb_loss / DC_loss / PSNR)  ->  compute_loss / summarize_loss (eikonal over mask_bg, 10^w weighted sum:
BA.py:186-218)  ->  loss.all.backward()  ->  Adam.step()  ->  ExponentialLR.step().

`surface_losses` is the point side of a BA iteration (BA.py:117-131, compute_loss "sfm" branch) on the fused point queries;
`render_losses` is that render-and-loss part for rays that are already picked (ray picking, poses, key points, COLMAP
bookkeeping are the drivers' camera-side logic: SURVEY section 2, out of scope); `RenderStage` adds the update and, with
`capture=True`, records the WHOLE step -- tracing kernel, point-query node of the traced depth, fused render with the loss
head inside, backward, Adam with the learning-rate schedule on the device -- into one hipGraph: a step is then a single
graph launch with no host synchronisation (the reference syncs at `mask_finish.sum() > 0`, at the tracing loop's
`.sum()` per trip and at `loss.item()`).
"""
from __future__ import annotations

import torch

from .losses import RenderLossHead, psnr
from .optim import FusedAdam


def render_losses(opt, renderer, sdf_field, rad_field, head, centers, rays, rgbs_gt, static_trips=False):
    """CameraSet.render(mode="train") after the ray pick + the render-side terms of compute_loss: -> the reference's `ret`
    keys (rgb, sdfs_volume, normals, depth_mlp, normal_mlp, mask_bg, rgb_loss, DC_loss, PSNR, tracing_loss) plus
    eikonal_loss (over mask_bg, BA.py:193-194), mse and `loss_all` (the head's 10^w weighted sum).
    centers, rays [B,R,3]; rgbs_gt [B,R,3].  The tracing runs first (it is independent of the render): its masks and depth
    are then inputs of the loss head that runs INSIDE the fused render (Renderer.forward_with_loss)."""
    b, r = centers.shape[:2]
    if static_trips:
        # one fused node forms the traced depth AND the two masks of Camera.py:515-516 (uint8, as the loss head takes them):
        # as torch ops these lines were ~25 launch-bound elementwise kernels of the captured step
        d_points, sdf_last, _, _ = sdf_field.sphere_tracing(centers.reshape(1, -1, 3), rays.reshape(1, -1, 3), sdf_field, iter=0,
                                                            static_trips=True, rgbs_gt=rgbs_gt.reshape(-1, 3))
        mask_bg8, mask_dc8 = sdf_field.last_masks
        mask_bg, mask_finish = mask_bg8.view(b, r), mask_dc8.view(b, r)
    else:
        d_points, sdf_last, _, mask_finish = sdf_field.sphere_tracing(centers.reshape(1, -1, 3), rays.reshape(1, -1, 3), sdf_field,
                                                                     iter=0)
        gray = rgbs_gt.mean(dim=-1)
        mask_bg = (gray < 0.95) & (gray > 0.05)                               # Camera.py:515
        mask_finish = mask_finish.view(b, r) & mask_bg                        # Camera.py:516
    ret, losses = renderer.forward_with_loss(opt, centers, rays, sdf_field, rad_field, head, rgbs_gt, d_points=d_points.view(b, r),
                                             mask_finish=mask_finish, mask_eik=mask_bg, mask_bg=mask_bg)
    if static_trips:
        mask_bg, mask_finish = mask_bg.view(torch.bool), mask_finish.view(torch.bool)      # 0 / 1 bytes: same storage, no kernel
    ret = dict(ret)
    ret.update(tracing_loss=0, mask_bg=mask_bg, mask_finish=mask_finish, d_points=d_points.view(b, r, 1),
               sdf_tracks=sdf_last.view(b, r, 1), rgb_loss=losses["rgb_loss"], DC_loss=losses["DC_loss"],
               eikonal_loss=losses["eikonal_loss"], mse=losses["mse"], PSNR=losses["PSNR"] if "PSNR" in losses else psnr(losses["mse"]),
               loss_all=losses["all"])
    return ret


def surface_losses(opt, sdf_field, xyzs, res=None):
    """The POINT side of a bundle-adjustment iteration (pipelines/BA.py:117-131) and the "sfm" branch of BA.compute_loss
    (BA.py:199-202): tracked 3-D points are projected onto the surface, `xyzs_new, normals_value = get_surface_pts(xyzs)`,
    re-evaluated, `sdfs = infer_sdf(xyzs_new)`, and give  sdf_surf = L1(sdfs, 0),  eikonal_loss = L1(normals_value, 1)
    and  mask_surf = |sdfs| < 2 * (extent / 10 / opt.Res).  Every field evaluation is one fused point-query node
    (ls2fm_sdf_eval forward, ls2fm_sdf_points_bwd backward with the analytic double backward of the normal); the
    re-projection term the drivers add on top of `xyzs_new` is camera-side logic (world2cam / cam2img), out of scope."""
    xyzs_new, normals_value = sdf_field.get_surface_pts(xyzs)
    sdfs = sdf_field.infer_sdf(xyzs_new, mode="ret_sdf").view(-1, 1)
    res = int(opt.Res) if res is None else int(res)
    sdf_threshold = (sdf_field.bound_max.reshape(-1)[0] - sdf_field.bound_min.reshape(-1)[0]) / 10 / res
    return dict(xyzs_new=xyzs_new, gradients=normals_value, sdfs=sdfs, mask_surf=sdfs.abs() < 2 * sdf_threshold,
                sdf_surf=sdfs.abs().mean(), eikonal_loss=(normals_value - 1).abs().mean())


class RenderStage:
    """render -> losses -> backward -> Adam + ExponentialLR over the two fields' parameters (and any extra ones, e.g. poses).

        stage = RenderStage(opt, renderer, sdf, rad, weights=opt.loss_weight.ba, lr=1e-2, lr_end=1e-4, max_iter=500)
        for it in range(500): ret = stage.step(centers, rays, rgbs_gt)

    capture=True: the first `step` call records the whole step into a hipGraph at the given batch shape; later calls copy
    the new rays into the captured input buffers and replay it (shapes must not change; no `.item()` anywhere).

    extra_loss: a callable `ret -> scalar tensor` evaluated inside the step and ADDED to `loss_all` before the backward -- the
    terms a driver forms outside the render (BA.run_ba's key-point re-projection error and, through `surface_losses`, its
    sdf_surf term: BA.py:119-147, 186-202), already weighted.  With capture=True it is recorded with the step: it must read
    its inputs from tensors that are updated in place and must not synchronise."""

    def __init__(self, opt, renderer, sdf_field, rad_field, weights=None, lr=1e-2, lr_end=1e-4, max_iter=1000, betas=(0.9, 0.999),
                 eps=1e-8, extra_params=(), capture=False, extra_loss=None):
        self.opt, self.renderer, self.sdf, self.rad = opt, renderer, sdf_field, rad_field
        dev = next(sdf_field.parameters()).device
        w = weights or {}
        get = (lambda k: w.get(k)) if isinstance(w, dict) else (lambda k: getattr(w, k, None))
        self.head = RenderLossHead(dev, w_rgb=get("rgb"), w_eikonal=get("eikonal_loss"), w_dc=get("DC_Loss"))
        self.params = [p for p in list(sdf_field.parameters()) + list(rad_field.parameters()) + list(extra_params) if p.requires_grad]
        self.gamma = (lr_end / lr) ** (1.0 / max_iter)                        # BA.py:87-88
        self.optim = FusedAdam(self.params, lr=lr, betas=betas, eps=eps, scheduled_gamma=self.gamma)
        self.capture = capture
        self.extra_loss = extra_loss
        self._graph = None
        self._one = torch.ones((), device=dev)

    def _eager(self, centers, rays, rgbs_gt, static_trips):
        for p in self.params:
            p.grad = None
        ret = render_losses(self.opt, self.renderer, self.sdf, self.rad, self.head, centers, rays, rgbs_gt, static_trips=static_trips)
        if self.extra_loss is not None:
            ret["loss_extra"] = self.extra_loss(ret)
            ret["loss_all"] = ret["loss_all"] + ret["loss_extra"]
        ret["loss_all"].backward(gradient=self._one)
        self.optim.step()
        return ret

    def step(self, centers, rays, rgbs_gt):
        if not self.capture:
            # the static form (trip count stays on the device, traced depth + masks as one fused node) whenever the fused tracing
            # kernel serves this field: no host round trip per step, ~40 fewer launches; else the reference-shaped form
            from . import fused as _fused
            static = _fused.available(self.sdf, centers)
            return self._eager(centers, rays, rgbs_gt, static_trips=static)
        if self._graph is None:
            from .graph import CapturedStep
            self._in = (centers.detach().clone(), rays.detach().clone(), rgbs_gt.detach().clone())
            # the capture warms the step up by running it for real: parameters, Adam state and the schedule are put back
            # afterwards (in place: the graph holds their addresses), so that this call, too, is exactly one step
            snap = self._snapshot()
            self._graph = CapturedStep(lambda: self._eager(*self._in, static_trips=True), params=self.params)
            self._restore(snap)
            out = self._graph.replay()
            self.optim.replayed(1)
            return out
        for dst, src in zip(self._in, (centers, rays, rgbs_gt)):
            if dst.shape != src.shape:
                raise RuntimeError("ls2fm.stage.RenderStage(capture=True): the batch shape is fixed by the first step")
            dst.copy_(src)
        out = self._graph.replay()
        self.optim.replayed(1)
        return out

    # ---- state snapshot around the capture's warm-up steps
    def _snapshot(self):
        opt = self.optim
        return dict(params=[p.detach().clone() for p in self.params],
                    state=[({k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state[p].items()} if p in opt.state else None)
                           for p in self.params],
                    lrs=[float(g["lr"]) for g in opt.param_groups],
                    sched={gi: t.clone() for gi, t in opt._sched.items()})

    def _restore(self, snap):
        opt = self.optim
        with torch.no_grad():
            for p, old in zip(self.params, snap["params"]):
                p.copy_(old)
                torch.autograd.graph.increment_version(p)
            for p, old in zip(self.params, snap["state"]):
                st = opt.state.get(p)
                if not st:
                    continue
                if old:
                    st["step"] = old["step"]
                    st["exp_avg"].copy_(old["exp_avg"])
                    st["exp_avg_sq"].copy_(old["exp_avg_sq"])
                else:                       # state created by the warm-up: back to its initial value
                    st["step"] = 0
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
            for g, lr in zip(opt.param_groups, snap["lrs"]):
                g["lr"] = lr
            for gi, t in opt._sched.items():
                if gi in snap["sched"]:
                    t.copy_(snap["sched"][gi])
                else:                       # schedule created by the warm-up: seeded from the step the group had BEFORE it
                    steps = [int(old["step"]) for p, old in zip(self.params, snap["state"])
                             if old and any(p is q for q in opt.param_groups[gi]["params"])]
                    t.copy_(torch.tensor([float(steps[0]) if steps else 0.0, snap["lrs"][gi], opt.scheduled_gamma, 0.0],
                                         dtype=torch.float64))
