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

import random

import torch

from .losses import RenderLossHead, psnr
from .optim import FusedAdam
from .utils import camera as _cam


_TRACE_STREAMS = {}


def _trace_stream(device, slot=0):
    """the stream(s) sphere tracings are launched on beside the render's forward: slot 0 the render rays', slots 1.. the key-point
    tracings of a loop's extra terms (latency-bound kernels of a few workgroups each: they overlap each other almost for free).
    None (= the current stream) inside a hipGraph capture under torch.distributed: the tracing max-reduces its trip count over the
    ranks on the stream it runs on, and a side-stream branch that forks again into RCCL's own stream is the depth-three fork tree
    that takes a capture down on ROCm 7.2 (csrc/streams.hip)."""
    from . import dist as _dist
    if _dist.is_distributed() and torch.cuda.is_current_stream_capturing():
        return None
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if (idx, slot) not in _TRACE_STREAMS:
        _TRACE_STREAMS[(idx, slot)] = torch.cuda.Stream(device=idx)
    return _TRACE_STREAMS[(idx, slot)]


class _EarlyTrace:
    """A key-point sphere tracing launched AHEAD of the render on a stream of its own (the render does not depend on it), its
    results picked up where the loss term is formed.  Holds the call's autograd-node record until the next launch: that record
    owns every tensor the other stream touches (no record_stream under hipGraph capture, see ls2fm.fused.TracedDepthNode)."""

    def __init__(self, sdf_field, slot):
        self.sdf, self.slot = sdf_field, slot
        self.out = self.ready = self.node = None

    def launch(self, center, ray):
        side = _trace_stream(center.device, self.slot)
        self.out = self.sdf.sphere_tracing(center, ray, self.sdf, static_trips=True, launch_stream=side)
        self.node = getattr(self.sdf, "last_trace_node", None)
        self.ready = torch.cuda.Event()
        self.ready.record(side)

    def take(self):
        """-> the tracing's outputs (joined into the current stream), or None when nothing was launched"""
        if self.out is None:
            return None
        torch.cuda.current_stream(self.out[0].device).wait_event(self.ready)
        out, self.out = self.out, None
        return out


def render_losses(opt, renderer, sdf_field, rad_field, head, centers, rays, rgbs_gt, static_trips=False, eikonal_over="bg"):
    """CameraSet.render(mode="train") after the ray pick + the render-side terms of compute_loss: -> the reference's `ret`
    keys (rgb, sdfs_volume, normals, depth_mlp, normal_mlp, mask_bg, rgb_loss, DC_loss, PSNR, tracing_loss) plus
    eikonal_loss (over mask_bg, BA.py:193-194), mse and `loss_all` (the head's 10^w weighted sum).
    centers, rays [B,R,3]; rgbs_gt [B,R,3].  The tracing runs first (it is independent of the render): its masks and depth
    are then inputs of the loss head that runs INSIDE the fused render (Renderer.forward_with_loss).
    eikonal_over: "bg" = over the rays of mask_bg (BA.py:193-194, Initialization.py:257-258), "all" = over every normal
    (rendering_refine.py:101-102)."""
    b, r = centers.shape[:2]
    ready = None
    if static_trips:
        # one fused node forms the traced depth AND the two masks of Camera.py:515-516 (uint8, as the loss head takes them):
        # as torch ops these lines were ~25 launch-bound elementwise kernels of the captured step.
        # The tracing (one latency-bound kernel of few workgroups, then the track evaluation) is independent of the render up to
        # the loss head inside shade_fwd: it runs on its own stream BESIDE the render's gather pass (L2-bound), and the forward
        # waits for its event only in front of shade_fwd.  (The autograd node itself belongs to the current stream -- its
        # backward runs there --; only the forward's kernels are launched on the other one.)
        side = _trace_stream(centers.device)
        d_points, sdf_last, _, _ = sdf_field.sphere_tracing(centers.reshape(1, -1, 3), rays.reshape(1, -1, 3), sdf_field, iter=0,
                                                            static_trips=True, rgbs_gt=rgbs_gt.reshape(-1, 3), launch_stream=side)
        mask_bg8, mask_dc8 = sdf_field.last_masks
        ready = torch.cuda.Event()
        ready.record(side)
        mask_bg, mask_finish = mask_bg8.view(b, r), mask_dc8.view(b, r)
        # the render's loss head evaluates mask_bg itself from the colours ("gt"): only the traced depth and mask_finish come out
        # of the tracing, and those are read by the loss REDUCTION behind the shading kernel -- the render waits for `ready` there
        head_bg = "gt"
    else:
        d_points, sdf_last, _, mask_finish = sdf_field.sphere_tracing(centers.reshape(1, -1, 3), rays.reshape(1, -1, 3), sdf_field,
                                                                     iter=0)
        gray = rgbs_gt.mean(dim=-1)
        mask_bg = (gray < 0.95) & (gray > 0.05)                               # Camera.py:515
        mask_finish = mask_finish.view(b, r) & mask_bg                        # Camera.py:516
        head_bg = mask_bg
    ret, losses = renderer.forward_with_loss(opt, centers, rays, sdf_field, rad_field, head, rgbs_gt, d_points=d_points.view(b, r),
                                             mask_finish=mask_finish, mask_eik=head_bg if eikonal_over == "bg" else None,
                                             mask_bg=head_bg, inputs_ready=ready,
                                             depth_node=getattr(sdf_field, "last_trace_node", None) if static_trips else None)
    if ready is not None:
        torch.cuda.current_stream(centers.device).wait_event(ready)       # join: later readers of the traced outputs, the backward
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
                 eps=1e-8, extra_params=(), capture=False, extra_loss=None, lr_color=None, eikonal_over="bg", reducer=None,
                 sharded=False, async_gather=False, extra_prepare=None, input_fn=None, share_gradients=False, shard_groups=1,
                 shard_views=False):
        """lr / lr_color: the reference's two field groups (`[{sdf_func.parameters(), lr_sdf}, {color_func.parameters(),
        lr_color}]`, BA.py:79-83; lr_color=None: one rate); extra_params: tensors (one more group at `lr`) or
        `{"params": [...], "lr": x}` dicts (the pose groups of BA.py:60-75).  ONE ExponentialLR factor for all groups,
        (lr_end / lr) ** (1 / max_iter), as the reference's scheduler (BA.py:87-88).
        reducer: an `ls2fm.dist.GradAllReducer` over the same parameters, called between backward and the update -- required
        when a process group is up (the fused loss head then divides by GLOBAL counts: without the reduction every rank would
        apply 1/world of its local gradient and the replicas would drift apart) -- or sharded=True: the update is
        `ls2fm.dist.ShardedAdam` (reduce-scatter of the flat gradient buffer, Adam on this rank's 1/world of the parameters,
        all-gather of the updated shards; async_gather: the all-gather runs on the communication stream until the next step's
        first parameter read; shard_groups >= 2: the exchange is pipelined by level groups of the two tables, `ShardedAdam`).
        Fields only: pose groups (extra_params) take the all-reduce form.
        reducer="auto": a GradAllReducer over this stage's parameters when a process group is up (else none).
        shard_views=True (with a process group of W ranks): the step's [B,R,3] batch is the GLOBAL one, identical on every rank,
        and rank r renders views r, r + W, ... of it (B % W == 0) -- BASELINE.json configs[3]: a BA step with the rays of 8
        registered views sharded over 8 GPUs; the loss head divides by global counts, the all-reduce sums the shards' gradients.
        `extra_loss` (terms formed outside the render: point side, re-projection, tracing consistency) is evaluated on every rank
        over the same inputs and enters the backward with weight 1 / W, so that the summed gradient is the single-process one;
        the returned `loss_all` is the global value on every rank.  capture=True under torch.distributed needs the RCCL ("nccl")
        backend: the step's collectives -- loss counts, trip count, gradient all-reduce -- are recorded into the hipGraph."""
        self.opt, self.renderer, self.sdf, self.rad = opt, renderer, sdf_field, rad_field
        dev = next(sdf_field.parameters()).device
        w = weights or {}
        get = (lambda k: w.get(k)) if isinstance(w, dict) else (lambda k: getattr(w, k, None))
        self.head = RenderLossHead(dev, w_rgb=get("rgb"), w_eikonal=get("eikonal_loss"), w_dc=get("DC_Loss"))
        groups = [dict(params=[p for p in sdf_field.parameters() if p.requires_grad], lr=lr),
                  dict(params=[p for p in rad_field.parameters() if p.requires_grad], lr=lr if lr_color is None else lr_color)]
        loose = [p for p in extra_params if not isinstance(p, dict)]
        if loose:
            groups.append(dict(params=[p for p in loose if p.requires_grad], lr=lr))
        groups += [dict(g) for g in extra_params if isinstance(g, dict)]
        groups = [g for g in groups if g["params"]]
        self.params = [p for g in groups for p in g["params"]]
        self.gamma = (lr_end / lr) ** (1.0 / max_iter)                        # BA.py:87-88
        self.sharded = bool(sharded)
        if self.sharded:
            if len(groups) > 2 or capture:
                raise NotImplementedError("ls2fm.stage.RenderStage(sharded=True): the two field groups only, eager steps only "
                                          "(pose groups and captured steps take the all-reduce form: reducer=\"auto\")")
            from .dist import ShardedAdam
            self.optim = ShardedAdam.for_fields(sdf_field, rad_field, lr=lr, lr_color=lr_color, betas=betas, eps=eps,
                                                scheduled_gamma=self.gamma, async_gather=async_gather, n_groups=shard_groups,
                                                in_backward=False)       # (a traced-depth node rides in every step)
            self.params = list(self.optim.params)
        else:
            self.optim = FusedAdam(groups, lr=lr, betas=betas, eps=eps, scheduled_gamma=self.gamma)
        self.capture = capture
        self.extra_loss = extra_loss
        # called at the start of a step, before the render is enqueued: work of the extra terms that does not depend on the
        # render (their key-point tracings) starts here, on streams of its own, and runs beside the render's forward
        self.extra_prepare = extra_prepare
        # `() -> (centers, rays, rgbs_gt)` evaluated INSIDE the step (and inside its capture): a loop's ray pick and pose algebra
        # from device tensors it updates in place -- `step()` then takes no arguments and a captured iteration has no eager
        # preamble (the loops' ~50 launch-bound kernels of camera arithmetic per iteration were half of their time)
        self.input_fn = input_fn
        # one gradient buffer per backward pass (ls2fm.fused.pass_gradient_sharing): only for a caller whose extra terms issue
        # ALL their field queries in `extra_prepare`, i.e. ahead of the render (the loops below)
        self.share_gradients = bool(share_gradients)
        self.eikonal_over = eikonal_over
        if isinstance(reducer, str):
            if reducer != "auto":
                raise ValueError(f"ls2fm.stage.RenderStage: reducer={reducer!r}")
            from . import dist as _dist
            reducer = _dist.GradAllReducer(self.params) if (_dist.is_distributed() and not self.sharded) else None
        self.reducer = reducer
        self.shard_views = bool(shard_views)
        self._graph = None
        self._table_versions = None
        self._one = torch.ones((), device=dev)

    def _eager(self, centers, rays, rgbs_gt, static_trips):
        from . import dist as _dist
        if self.sharded:
            self.optim.wait_params()               # the previous step's all-gather (async_gather) before anything reads a parameter
        if _dist.is_distributed() and self.reducer is None and not self.sharded:
            raise RuntimeError("ls2fm.stage.RenderStage under torch.distributed needs reducer=GradAllReducer(stage.params): the loss "
                               "head normalises by global counts, the gradients must be summed over the ranks before the update")
        for p in self.params:
            p.grad = None
        world = 1
        if self.shard_views and _dist.is_distributed():
            import torch.distributed as tdist
            world, rank = tdist.get_world_size(), tdist.get_rank()
            if centers.shape[0] % world:
                raise RuntimeError(f"ls2fm.stage.RenderStage(shard_views=True): {centers.shape[0]} views over {world} ranks")
            if world > 1:                  # views rank, rank + world, ...: strided views of the global batch, made contiguous
                centers, rays, rgbs_gt = (t[rank::world].contiguous() for t in (centers, rays, rgbs_gt))
        if self.extra_prepare is not None and static_trips:
            self.extra_prepare()
        ret = render_losses(self.opt, self.renderer, self.sdf, self.rad, self.head, centers, rays, rgbs_gt, static_trips=static_trips,
                            eikonal_over=self.eikonal_over)
        loss_bwd = ret["loss_all"]
        if self.extra_loss is not None:
            ret["loss_extra"] = self.extra_loss(ret)
            # (replicated on every rank of a view-sharded step: 1 / world of it per rank in the backward, the whole in the log)
            total = ret["loss_all"] + ret["loss_extra"]
            loss_bwd = total if world == 1 else ret["loss_all"] + ret["loss_extra"] * (1.0 / world)
            ret["loss_all"] = total
        from . import fused as _fused
        with _fused.pass_gradient_sharing(self.share_gradients and static_trips and self.extra_prepare is not None):
            loss_bwd.backward(gradient=self._one)
        if self.reducer is not None:
            self.reducer.all_reduce()
        self.optim.step()
        return ret

    def step(self, centers=None, rays=None, rgbs_gt=None):
        if centers is None:
            if self.input_fn is None:
                raise TypeError("ls2fm.stage.RenderStage.step(): centers, rays, rgbs_gt -- or construct the stage with input_fn")
            if not self.capture:
                centers, rays, rgbs_gt = self.input_fn()
            elif self._graph is None:
                return self._capture(lambda: self._eager(*self.input_fn(), static_trips=True))
            else:
                return self._replay()
        if not self.capture:
            # the static form (trip count stays on the device, traced depth + masks as one fused node) whenever the fused tracing
            # kernel serves this field: no host round trip per step, ~40 fewer launches; else the reference-shaped form
            from . import fused as _fused
            static = _fused.available(self.sdf, centers)
            return self._eager(centers, rays, rgbs_gt, static_trips=static)
        if self._graph is None:
            self._in = (centers.detach().clone(), rays.detach().clone(), rgbs_gt.detach().clone())
            return self._capture(lambda: self._eager(*self._in, static_trips=True))
        srcs = (centers, rays, rgbs_gt)
        for dst, src in zip(self._in, srcs):
            if dst.shape != src.shape:
                raise RuntimeError("ls2fm.stage.RenderStage(capture=True): the batch shape is fixed by the first step")
        torch._foreach_copy_(list(self._in), [s.detach() for s in srcs])         # one launch for the three inputs
        return self._replay()

    def _replay(self):
        """one replay of the captured step.  The graph reads the entry-interleaved table copy WITHOUT checking it (inside a capture
        nothing can be checked): a foreign writer of a table since the last replay -- a checkpoint load, `load_state_dict`, another
        optimizer -- is detected here, on the host, through the tables' version counters, and costs one rebuild of the copy."""
        from . import fused as _fused
        tabs = _fused.table_params(self.sdf, self.rad)
        if tabs is not None and [t._version for t in tabs] != self._table_versions:
            _fused.sync_mirror(self.sdf, self.rad)
        out = self._graph.replay()
        self.optim.replayed(1)
        self._table_versions = None if tabs is None else [t._version for t in tabs]
        return out

    # ---- checkpoint / resume (utils/util.py:198-259 stores `optim_*` / `sched_*` state dicts next to the fields')
    def state_dict(self):
        """the optimizer's state in torch's layout (per-parameter `step, exp_avg, exp_avg_sq`, per-group `lr`): with the device-
        resident schedule that IS the scheduler state -- ExponentialLR's only state is the current rate"""
        if self.sharded:
            raise NotImplementedError("ls2fm.stage.RenderStage.state_dict: use ShardedAdam.state_dict() (per-rank shards)")
        return {"optim": self.optim.state_dict(), "gamma": self.gamma}

    def load_state_dict(self, state):
        """resume: moments, step counts and rates are copied INTO the tensors this stage (and a captured graph of it) already
        owns; the next step -- eager or a replay -- continues the saved trajectory.  Load the fields' state dicts first."""
        self.optim.load_state_dict(state["optim"])
        self._table_versions = None            # the next replay re-checks the interleaved copy against the (reloaded) tables

    def _capture(self, fn):
        from . import dist as _dist
        if _dist.is_distributed():
            import torch.distributed as tdist
            if tdist.get_backend() != "nccl" or self.sharded:
                raise NotImplementedError("ls2fm.stage.RenderStage(capture=True) under torch.distributed needs the RCCL (\"nccl\") "
                                          "backend and the all-reduce form (reducer=...): its collectives -- loss counts, trip count, "
                                          "gradient all-reduce -- are then recorded into the hipGraph; gloo moves tensors through "
                                          "the host, and the sharded update keeps per-rank state the snapshot around a capture does "
                                          "not cover.  Use capture=False")
        from .graph import CapturedStep
        # the capture warms the step up by running it for real: parameters, Adam state and the schedule are put back
        # afterwards (in place: the graph holds their addresses), so that this call, too, is exactly one step
        snap = self._snapshot()
        # the captured step's own Adam keeps the interleaved table copy current (ls2fm.fused): no rebuild inside the graph
        from . import fused as _fused
        _fused.trust_mirror_in_capture(self.sdf, self.rad)
        self._graph = CapturedStep(fn, params=self.params)
        self._restore(snap)
        self._table_versions = None                     # the restore rewrote the tables behind the graph's back: _replay re-syncs
        return self._replay()

    # ---- state snapshot around the capture's warm-up steps
    def _snapshot(self):
        return _snapshot_training_state(self.params, self.optim)

    def _restore(self, snap):
        _restore_training_state(self.params, self.optim, snap)


def _snapshot_training_state(params, opt):
    """parameters, Adam moments / step counts, learning rates and device schedules -- what a capture's warm-up steps change"""
    return dict(params=[p.detach().clone() for p in params],
                state=[({k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state[p].items()} if p in opt.state else None)
                       for p in params],
                lrs=[float(g["lr"]) for g in opt.param_groups],
                sched={gi: t.clone() for gi, t in opt._sched.items()})


def _restore_training_state(params, opt, snap):
    """put a `_snapshot_training_state` back IN PLACE (a captured graph holds the addresses)"""
    with torch.no_grad():
        for p, old in zip(params, snap["params"]):
            p.copy_(old)
            torch.autograd.graph.increment_version(p)
        for p, old in zip(params, snap["state"]):
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
                steps = [int(old["step"]) for p, old in zip(params, snap["state"])
                         if old and any(p is q for q in opt.param_groups[gi]["params"])]
                t.copy_(torch.tensor([float(steps[0]) if steps else 0.0, snap["lrs"][gi], opt.scheduled_gamma, 0.0],
                                     dtype=torch.float64))


# ================================================================================================ the stage LOOPS
# `Refine.run` (pipelines/rendering_refine.py:72-97) and `BA.run_ba` (pipelines/BA.py:110-188) as loops over `RenderStage`:
# every iteration is the reference's -- ray pick, multi-view tracing consistency of one random camera's key points, render +
# tracing + masks + losses, (BA: the point side and the re-projection through the pose parameters), one backward, one Adam
# update over every parameter group, one ExponentialLR step -- with nothing read back to the host inside an iteration: the
# reference's `.item()` (the adaptive re-projection weight, BA.py:164), `mask_surf.sum() == 0` and `mask_finish.sum() > 0`
# branches are device-side selects here, the logs come back as device tensors.  Scene containers are plain tensors: the
# reference's Camera / Point3D bookkeeping (COLMAP ids, feature tracks) stays with the caller.
class TrackedViews:
    """What the loops need of a CameraSet + Point3DSet: world-to-camera poses [V,3,4] (or se(3) parameters [V,6]), one
    intrinsic matrix [3,3], the images as [V, H*W, 3], per view its key points [n_v, 2] (pixels) and the index of the tracked 3-D
    point each of them observes [n_v] (Camera.idx2d_to_3d entries != -1), the points [M,3]."""

    def __init__(self, poses, intrinsic, images, keypoints, track_ids, xyzs, H, W):
        self.poses, self.intrinsic, self.images = poses, intrinsic, images
        self.keypoints, self.track_ids, self.xyzs = list(keypoints), [t.long() for t in track_ids], xyzs
        self.H, self.W = int(H), int(W)
        self.grid = _cam.mesh_grid(H=H, W=W, device=images.device)
        n_max = max(k.shape[0] for k in self.keypoints)
        dev = images.device
        # key points padded to one shape (a captured step has fixed shapes): padding repeats the view's first key point -- a
        # duplicate ray finishes its sphere tracing exactly when the original does, so the global trip count is unchanged
        self.kp_pad = torch.stack([torch.cat([k, k[:1].expand(n_max - k.shape[0], 2)]) for k in self.keypoints]).to(dev)
        self.id_pad = torch.stack([torch.cat([t, t[:1].expand(n_max - t.shape[0])]) for t in self.track_ids]).to(dev)
        self.kp_live = torch.stack([torch.arange(n_max, device=dev) < k.shape[0] for k in self.keypoints]).float()
        # K^-1 on the host for the fused ray construction (ls2fm_camera_rays): read back here, once, outside any captured step
        self.kinv_host = _cam.host_inverse_intrinsic(intrinsic) if images.is_cuda else None


def keypoint_rays(pose, intrinsic, kypts):
    """centers / rays of a view's key points as `Camera.get_pts3D` forms them (pipelines/Camera.py:129-133): pose [3,4],
    kypts [n,2] -> [1,n,3] each"""
    in_cam = _cam.img2cam(_cam.to_hom(kypts), intrinsic.unsqueeze(0))            # [1,n,3]
    center = _cam.cam2world(torch.zeros_like(in_cam), pose.unsqueeze(0))
    return center, _cam.cam2world(in_cam, pose.unsqueeze(0)) - center


class TracingConsistency:
    """The multi-view tracing-consistency block of `CameraSet.render` (pipelines/Camera.py:466-476) as a step's extra loss:
    the key points of ONE view are sphere-traced onto the surface; tracing_loss = mean ||xyz - surface point||, and the SDF at
    the end of the tracks becomes `ret.sdfs` (sdf_surf = mean |sdfs|).  The view's rays live in buffers that `select()` rewrites
    in place, so a captured step sees the new view at replay."""

    def __init__(self, sdf_field, views, w_tracing, w_surf, static_trips=True):
        self.sdf, self.views, self.static = sdf_field, views, static_trips
        self.w_tracing = 0.0 if w_tracing is None else 10.0 ** float(w_tracing)
        self.w_surf = 0.0 if w_surf is None else 10.0 ** float(w_surf)
        n = views.kp_pad.shape[1]
        dev = views.kp_pad.device
        self.center, self.ray = torch.zeros(1, n, 3, device=dev), torch.zeros(1, n, 3, device=dev)
        self.target, self.live = torch.zeros(n, 3, device=dev), torch.zeros(n, device=dev)
        self.use_sdfs = True                   # False: `ret.sdfs` was set by the caller before the render (BA.py:122, Camera.py:474)
        self._early = _EarlyTrace(sdf_field, slot=1)

    def prepare(self):
        """launch the tracing now (RenderStage(extra_prepare=...)): it runs beside the render's forward"""
        if self.static:
            self._early.launch(self.center, self.ray)

    @torch.no_grad()
    def select(self, view, poses, fixed=None):
        """view: a Python int, or a DEVICE long tensor [1] (then nothing here touches the host: the selection can sit inside a
        captured step and follow the tensor's value at every replay).  fixed: a `_FixedPoseRays` of these poses."""
        if torch.is_tensor(view):
            pick = lambda t: t.index_select(0, view)[0]

            def into(src, index, out):
                # `out=` neither casts nor broadcasts, and a shape mismatch makes torch RESIZE the out tensor -- new storage, while
                # a captured graph and the early tracing hold the old address (round-4 advisor): refuse instead
                want = (index.numel(),) + tuple(src.shape[1:])
                if src.dtype != out.dtype or tuple(out.shape) != want:
                    raise RuntimeError(f"ls2fm.stage.TracingConsistency.select: source {tuple(src.shape)} {src.dtype} does not "
                                       f"fit the persistent buffer {tuple(out.shape)} {out.dtype}")
                torch.index_select(src, 0, index, out=out)
            if fixed is not None:
                into(fixed.kp_center, view, self.center)
                into(fixed.kp_ray, view, self.ray)
            elif self.views.kinv_host is not None and poses.shape[-1] == 4:
                # one launch: the selected view's pose and key points -> its rays, written into the buffers the tracing reads
                _cam.camera_rays(self.views.kinv_host, poses=poses, xy=self.views.kp_pad, view_sel=view, out=(self.center, self.ray))
            else:
                c, r = keypoint_rays(pick(poses), self.views.intrinsic, pick(self.views.kp_pad))
                self.center.copy_(c); self.ray.copy_(r)
            # index_select straight INTO the persistent buffers: a `.copy_()` of a fresh result is a device-to-device memcpy node in
            # a captured iteration -- 20 - 80 us each on this stack, against 4 us for the gather kernel writing in place
            into(self.views.xyzs, pick(self.views.id_pad), self.target)
            into(self.views.kp_live, view, self.live.view(1, -1))
            return
        c, r = keypoint_rays(poses[view], self.views.intrinsic, self.views.kp_pad[view])
        self.center.copy_(c); self.ray.copy_(r)
        self.target.copy_(self.views.xyzs[self.views.id_pad[view]])
        self.live.copy_(self.views.kp_live[view])

    def __call__(self, ret, raw=False):
        """raw=True (fused CUDA path without sdf_surf only): the UNWEIGHTED tracing loss -- the caller multiplies by `w_tracing`"""
        early = self._early.take()
        d, sdf_last = early[:2] if early is not None else self.sdf.sphere_tracing(self.center, self.ray, self.sdf, static_trips=self.static)[:2]
        if d.is_cuda and d.dtype == torch.float32:
            # one node each way (ls2fm_tracing_term_fwd / _bwd): in a captured iteration every torch kernel of the lines below and
            # of their autograd mirror is a graph node of >= 4.6 us, whatever its size
            terms = _TracingTerm.apply(self.center, self.ray, d, self.target, self.live, sdf_last if self.use_sdfs else None)
            ret["tracing_loss"] = terms[0]
            if raw and not self.use_sdfs:          # the caller weights it inside a fused node of its own (BALoop)
                return ret["tracing_loss"]
            if self.use_sdfs:
                ret["sdf_surf"] = terms[1]
                return _weighted_pair(ret["tracing_loss"], ret["sdf_surf"], self.w_tracing, self.w_surf)
            return self.w_tracing * ret["tracing_loss"]
        # (few, fat torch ops: in a captured iteration every elementwise kernel here and in its backward is ~8 us of launch gap)
        surface = torch.addcmul(self.center[0], self.ray[0], d.reshape(-1, 1))
        weight = self.live / self.live.sum()                                    # no graph: 1 / count on the live key points
        ret["tracing_loss"] = torch.dot(torch.linalg.vector_norm(self.target - surface, dim=-1), weight)
        loss = self.w_tracing * ret["tracing_loss"]
        if self.use_sdfs:
            ret["sdf_surf"] = torch.dot(sdf_last.reshape(-1).abs(), weight)
            loss = loss + self.w_surf * ret["sdf_surf"]
        return loss


class _TracingTerm(torch.autograd.Function):
    """the tracing-consistency sums of TracingConsistency.__call__ as one fused node (include/ls2fm.h: ls2fm_tracing_term_fwd /
    _bwd): (tracing_loss, sdf_surf) from the traced depths d [.., n] and, optionally, the last SDF values; differentiable w.r.t.
    both"""

    @staticmethod
    def forward(ctx, center, ray, d, target, live, sdf_last):
        from . import _lib
        lib = _lib.load()
        n = live.numel()
        c, r, t = center.detach().reshape(-1, 3).contiguous(), ray.detach().reshape(-1, 3).contiguous(), target.detach().contiguous()
        dd, lv = d.detach().reshape(-1).contiguous(), live.detach().contiguous()
        sl = None if sdf_last is None else sdf_last.detach().reshape(-1).contiguous()
        out = torch.empty(3, device=d.device)
        _lib.check(lib.ls2fm_tracing_term_fwd(_lib.ptr(c), _lib.ptr(r), _lib.ptr(dd), _lib.ptr(t), _lib.ptr(lv), _lib.ptr(sl), n,
                                              _lib.ptr(out), _lib.stream_ptr()), "ls2fm_tracing_term_fwd")
        ctx.save_for_backward(c, r, dd, t, lv, out, *([] if sl is None else [sl]))
        ctx.shapes = (d.shape, None if sdf_last is None else sdf_last.shape)
        # the two terms as tensors of their own (views of `out` made HERE, outside autograd): indexing a stacked result afterwards
        # costs a zero fill + a copy per term in the backward -- graph nodes of a captured iteration
        ctx.set_materialize_grads(False)
        return out[0], out[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_tl, g_sd):
        from . import _lib
        lib = _lib.load()
        c, r, dd, t, lv, out = ctx.saved_tensors[:6]
        sl = ctx.saved_tensors[6] if len(ctx.saved_tensors) > 6 else None
        g_tl = None if g_tl is None else g_tl.reshape(1).float().contiguous()
        g_sd = None if g_sd is None else g_sd.reshape(1).float().contiguous()
        d_d = torch.empty_like(dd)
        d_s = None if sl is None else torch.empty_like(sl)
        _lib.check(lib.ls2fm_tracing_term_bwd(_lib.ptr(c), _lib.ptr(r), _lib.ptr(dd), _lib.ptr(t), _lib.ptr(lv), _lib.ptr(sl), lv.numel(),
                                              _lib.ptr(out), _lib.ptr(g_tl), _lib.ptr(g_sd), _lib.ptr(d_d), _lib.ptr(d_s), _lib.stream_ptr()),
                   "ls2fm_tracing_term_bwd")
        return None, None, d_d.view(ctx.shapes[0]), None, None, (None if d_s is None else d_s.view(ctx.shapes[1]))


class _WeightedPair(torch.autograd.Function):
    """wa a + wb b of two device scalars as one node each way (ls2fm_weighted_pair_fwd / _bwd).  The two gradients are the halves
    of ONE two-float buffer: a producer that returned `a` and `b` as the halves of one buffer too (_MatchTerm) gets its upstream
    back contiguous and needs no stacking kernel"""

    @staticmethod
    def forward(ctx, a, b, wa, wb):
        from . import _lib
        lib = _lib.load()
        out = torch.empty((), device=a.device)
        _lib.check(lib.ls2fm_weighted_pair_fwd(_lib.ptr(a.detach()), _lib.ptr(b.detach()), float(wa), float(wb), _lib.ptr(out),
                                               _lib.stream_ptr()), "ls2fm_weighted_pair_fwd")
        ctx.w = (float(wa), float(wb), a.shape, b.shape)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        wa, wb, sa, sb = ctx.w
        d2 = torch.empty(2, device=g.device)
        _lib.check(lib.ls2fm_weighted_pair_bwd(_lib.ptr(g.reshape(1).float().contiguous()), wa, wb, _lib.ptr(d2), _lib.stream_ptr()),
                   "ls2fm_weighted_pair_bwd")
        return d2[0].view(sa), d2[1].view(sb), None, None


def _weighted_pair(a, b, wa, wb):
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.numel() == 1 and b.numel() == 1:
        return _WeightedPair.apply(a, b, wa, wb)
    return wa * a + wb * b


class _BATerms(torch.autograd.Function):
    """BA.run_ba's loss lines outside the render (pipelines/BA.py:160-170) as one fused node each way (include/ls2fm.h:
    ls2fm_ba_terms_fwd / _bwd): (sdf_surf, w_reproj, extra) with sdf_surf = mean |sdfs|, w_reproj = w_hi where the re-projection
    error exceeds `thresh` else w_lo (from the detached error), extra = w_reproj reproj + w_surf sdf_surf + w_add add.  Only
    `extra` carries a gradient (the other two are the log's)."""

    @staticmethod
    def forward(ctx, reproj, sdfs, add, thresh, w_lo, w_hi, w_surf, w_add):
        from . import _lib
        lib = _lib.load()
        r = reproj.detach().reshape(1).float().contiguous()
        s = sdfs.detach().reshape(-1).float().contiguous()
        a = None if add is None else add.detach().reshape(1).float().contiguous()
        surf, w, extra = (torch.empty((), device=s.device) for _ in range(3))
        _lib.check(lib.ls2fm_ba_terms_fwd(_lib.ptr(r), _lib.ptr(s), s.numel(), _lib.ptr(a), float(thresh), float(w_lo), float(w_hi),
                                          float(w_surf), float(w_add), _lib.ptr(surf), _lib.ptr(w), _lib.ptr(extra), _lib.stream_ptr()),
                   "ls2fm_ba_terms_fwd")
        ctx.save_for_backward(s, w)
        ctx.consts = (float(w_surf), float(w_add), reproj.shape, sdfs.shape, None if add is None else add.shape)
        ctx.mark_non_differentiable(surf, w)
        ctx.set_materialize_grads(False)
        return surf, w, extra

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_surf, g_w, g):
        from . import _lib
        lib = _lib.load()
        s, w = ctx.saved_tensors
        w_surf, w_add, shape_r, shape_s, shape_a = ctx.consts
        if g is None:
            return (None,) * 8
        g = g.reshape(1).float().contiguous()
        d_r, d_s = torch.empty(1, device=s.device), torch.empty_like(s)
        d_a = None if shape_a is None else torch.empty(1, device=s.device)
        _lib.check(lib.ls2fm_ba_terms_bwd(_lib.ptr(s), s.numel(), _lib.ptr(w), _lib.ptr(g), w_surf, w_add, _lib.ptr(d_r), _lib.ptr(d_s),
                                          _lib.ptr(d_a), _lib.stream_ptr()), "ls2fm_ba_terms_bwd")
        return d_r.view(shape_r), d_s.view(shape_s), (None if d_a is None else d_a.view(shape_a)), None, None, None, None, None


class _MatchTerm(torch.autograd.Function):
    """the explicit-match terms of the two-view initialisation as one fused node (include/ls2fm.h: ls2fm_match_term_fwd / _bwd):
    (reproj_error, sdf_surf) from the traced depths and last SDF values of each source view; `surface` receives the traced points"""

    @staticmethod
    def forward(ctx, fixed, surface, *ds):
        import ctypes
        from . import _lib
        lib = _lib.load()
        center, ray, uv_obs, poses, k_host, n = fixed
        S = len(ds) // 2
        d = [t.detach().reshape(-1).float().contiguous() for t in ds[:S]]
        sl = [t.detach().reshape(-1).float().contiguous() for t in ds[S:]]
        arr = lambda ts: (ctypes.c_void_p * S)(*[t.data_ptr() for t in ts])          # noqa: E731
        out = torch.empty(2, device=center.device)
        _lib.check(lib.ls2fm_match_term_fwd(_lib.ptr(center), _lib.ptr(ray), _lib.ptr(uv_obs), _lib.ptr(poses), k_host, S, n, arr(d), arr(sl),
                                            _lib.ptr(surface), _lib.ptr(out), _lib.stream_ptr()), "ls2fm_match_term_fwd")
        ctx.fixed, ctx.S = fixed, S
        ctx.shapes = [t.shape for t in ds]
        ctx.save_for_backward(*d, *sl)
        ctx.set_materialize_grads(False)
        return out[0], out[1]            # (tensors of their own: no select backward -- a fill + a copy per term -- behind them)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g0, g1):
        import ctypes
        from . import _lib
        lib = _lib.load()
        center, ray, uv_obs, poses, k_host, n = ctx.fixed
        S = ctx.S
        saved = ctx.saved_tensors
        d, sl = list(saved[:S]), list(saved[S:])
        dd, dsl = [torch.empty_like(t) for t in d], [torch.empty_like(t) for t in sl]
        arr = lambda ts: (ctypes.c_void_p * S)(*[t.data_ptr() for t in ts])          # noqa: E731
        if g0 is not None and g1 is not None and g0.dtype == torch.float32 and g1.dtype == torch.float32 and \
                g1.data_ptr() == g0.data_ptr() + 4:
            g = g0                       # the halves of one buffer (_WeightedPair's backward): the kernel reads g[0], g[1] from there
        else:
            zero = torch.zeros((), device=center.device)
            g = torch.stack([(zero if t is None else t).reshape(()).float() for t in (g0, g1)])
        _lib.check(lib.ls2fm_match_term_bwd(_lib.ptr(center), _lib.ptr(ray), _lib.ptr(uv_obs), _lib.ptr(poses), k_host, S, n, arr(d), arr(sl),
                                            _lib.ptr(g), arr(dd), arr(dsl), _lib.stream_ptr()), "ls2fm_match_term_bwd")
        grads = [t.view(sh) for t, sh in zip(dd + dsl, ctx.shapes)]
        return (None, None, *grads)


class _HostPicks:
    """The per-iteration ray pick of the loops -- the first k entries of a random permutation of the H W pixels, as the reference's
    `torch.randperm(H * W, device=...)[:k]` (pipelines/Camera.py:263, 431, 468) -- drawn on the HOST (numpy `Generator.choice`
    without replacement: the same distribution; seeded from `torch.initial_seed()`) and handed to the device with ONE asynchronous
    copy, together with the iteration's view index, from a ring of pinned buffers.  The device-side randperm is a radix sort of
    H W keys: 12 launches / 56 us in front of every captured iteration, for 15 us of host time that overlaps the previous
    iteration.  `dev[:k]` are the pixel indices, `dev[k:]` the tail values (the view); `device_picks=True` on a loop keeps the
    device-side draw."""
    RING = 8
    _instances = 0          # every loop instance draws from a stream of its own (round-5 advisor: they all replayed ONE sequence)

    def __init__(self, n_total, k, device, tail=1, shared=False):
        """shared=True (a loop under torch.distributed: every rank must evaluate the replicated terms on the SAME pixels and view):
        the generator is seeded with rank 0's seed -- one 8-byte broadcast at construction, no collective per iteration -- and
        the view index comes from the same generator (`view(V)`) instead of the process-global, per-rank `random` module."""
        import numpy as np
        self.n, self.k, self.tail = int(n_total), int(k), int(tail)
        seed = (int(torch.initial_seed()) + 0x9E3779B97F4A7C15 * _HostPicks._instances) & 0x7FFFFFFFFFFFFFFF
        _HostPicks._instances += 1
        from . import dist as _dist
        self.shared = bool(shared) and _dist.is_distributed()
        if self.shared:
            import torch.distributed as tdist
            t = torch.tensor([seed], dtype=torch.long, device=device if tdist.get_backend() == "nccl" else "cpu")
            tdist.broadcast(t, 0)
            seed = int(t.item())
        self.rng = np.random.default_rng(seed)
        self.dev = torch.zeros(self.k + self.tail, dtype=torch.long, device=device)
        cuda = self.dev.is_cuda
        self.host = [torch.zeros(self.k + self.tail, dtype=torch.long).pin_memory() if cuda else torch.zeros(self.k + self.tail, dtype=torch.long)
                     for _ in range(self.RING if cuda else 1)]
        self.events = [None] * len(self.host)
        self.slot = 0

    def view(self, n_views):
        """the iteration's random view (the reference: `random.randint(0, V - 1)`): from this generator when the picks are shared
        between ranks, else from the `random` module as the reference draws it"""
        return int(self.rng.integers(0, n_views)) if self.shared else random.randint(0, n_views - 1)

    def shared_pixels(self, rays_idx):
        """a device-side pixel draw under a shared pick: rank 0's"""
        if self.shared:
            import torch.distributed as tdist
            tdist.broadcast(rays_idx, 0)
        return rays_idx

    def draw(self, *tail_values):
        h, ev = self.host[self.slot], self.events[self.slot]
        if ev is not None:
            ev.synchronize()                     # (the copy that last read this pinned buffer: RING iterations ago)
        h[:self.k] = torch.from_numpy(self.rng.choice(self.n, self.k, replace=False))
        for q, v in enumerate(tail_values):
            h[self.k + q] = int(v)
        self.dev.copy_(h, non_blocking=True)
        if self.dev.is_cuda:
            self.events[self.slot] = torch.cuda.Event()
            self.events[self.slot].record()
        self.slot = (self.slot + 1) % len(self.host)


def _pick_rays(views, poses, rays_idx, se3=None, poses_out=None):
    """CameraSet.render's ray pick for given poses (Camera.py:457-463): the same pixels in every view.  se3 [V,6]: the poses are
    the exponentials of these parameters, formed in the same launch (and left in poses_out)"""
    if views.kinv_host is not None and not (poses if se3 is None else se3).requires_grad:
        centers, rays = _cam.camera_rays(views.kinv_host, poses=None if se3 is not None else poses, se3=se3, pix=rays_idx, width=views.W,
                                         poses_out=poses_out)
    else:
        if se3 is not None:
            poses = _cam.lie.se3_to_SE3(se3)
            if poses_out is not None:
                poses_out.copy_(poses)
        centers, rays = _cam.get_center_and_ray(None, poses, intr=views.intrinsic.unsqueeze(0), rays_idx=rays_idx, xy_grid=views.grid)
    return centers, rays, views.images[:, rays_idx, :]


class _FixedPoseRays:
    """Loops whose poses do not move (Refine, Init): the rays of EVERY pixel and of every (padded) key point are formed once;
    an iteration then only gathers its pick -- three index kernels instead of the ~40 launch-bound ones of the pinhole / pose
    algebra (which cost a captured iteration 0.3 ms).  Same arithmetic as `_pick_rays` / `keypoint_rays`: a ray does not depend on
    which other pixels are picked with it."""

    def __init__(self, views, poses):
        with torch.no_grad():
            self.centers, self.rays = _cam.get_center_and_ray(None, poses, intr=views.intrinsic.unsqueeze(0), rays_idx=None, xy_grid=views.grid)
            kc, kr = zip(*[keypoint_rays(poses[v], views.intrinsic, views.kp_pad[v]) for v in range(poses.shape[0])])
            self.kp_center, self.kp_ray = torch.cat(kc, dim=0), torch.cat(kr, dim=0)              # [V, n_max, 3]
        self.views = views

    def pick(self, rays_idx):
        return self.centers[:, rays_idx, :], self.rays[:, rays_idx, :], self.views.images[:, rays_idx, :]


class RefineLoop:
    """`Refine` (pipelines/rendering_refine.py:15-126): fields only, fixed poses, eikonal over every normal, tracing consistency
    and sdf_surf on one random view's key points.

        loop = RefineLoop(opt, renderer, sdf, rad, views, weights=opt.loss_weight.refine, lr_sdf=1e-3, lr_sdf_end=5e-4,
                          lr_color=1e-3, max_iter=500, rand_rays=8192)
        logs = loop.run()                      # {"all": [max_iter], "PSNR": ..., ...} device tensors

    picks: optional per-iteration (rays_idx, view) pairs (parity tests replay the reference's draws); default: the pixel indices and
    the view drawn on the host and copied in one asynchronous transfer (`_HostPicks`; `device_picks=True`: the reference's
    device-side `torch.randperm(H * W)` head) -- no device synchronisation either way."""

    def __init__(self, opt, renderer, sdf_field, rad_field, views, weights, lr_sdf, lr_sdf_end, lr_color, max_iter, rand_rays,
                 capture=False, static_trips=None, distributed=False, device_picks=False):
        """distributed=True (a process group is up; every rank holds the same views, weights and per-iteration picks): the render
        rays are sharded by view, the gradients all-reduced (RenderStage(shard_views=True, reducer="auto"))"""
        get = (lambda k: weights.get(k)) if isinstance(weights, dict) else (lambda k: getattr(weights, k, None))
        self.views, self.max_iter, self.rand_rays = views, int(max_iter), int(rand_rays)
        from . import fused as _fused
        static = _fused.available(sdf_field, views.images) if static_trips is None else static_trips
        self.extra = TracingConsistency(sdf_field, views, get("tracing_loss"), get("sdf_surf"), static_trips=static)
        self.stage = RenderStage(opt, renderer, sdf_field, rad_field, weights=weights, lr=lr_sdf, lr_end=lr_sdf_end, max_iter=max_iter,
                                 lr_color=lr_color, capture=capture, extra_loss=self.extra, eikonal_over="all", extra_prepare=self.extra.prepare,
                                 input_fn=self._inputs, share_gradients=bool(static), shard_views=bool(distributed),
                                 reducer="auto" if distributed else None)
        self.poses = views.poses if views.poses.shape[-1] == 4 else _cam.lie.se3_to_SE3(views.poses)
        self._keys = ("loss_all", "PSNR", "rgb_loss", "DC_loss", "eikonal_loss", "sdf_surf", "tracing_loss")
        # an iteration's picks as device tensors, updated in place: the ray pick and the key-point rays are formed INSIDE the step
        dev = self.poses.device
        k = self.rand_rays // self.poses.shape[0]
        self._picks = _HostPicks(views.H * views.W, k, dev, tail=1, shared=distributed)   # pixel indices + the view, one buffer, one copy per iteration
        self._idx, self._view = self._picks.dev[:k], self._picks.dev[k:]
        self.device_picks = bool(device_picks)
        self._fixed = _FixedPoseRays(views, self.poses)

    def _inputs(self):
        self.extra.select(self._view, self.poses, fixed=self._fixed)
        return self._fixed.pick(self._idx)

    def step(self, rays_idx=None, view=None):
        V = self.poses.shape[0]
        if rays_idx is None and view is None and not self.device_picks:
            self._picks.draw(self._picks.view(V))
            return self.stage.step()
        if rays_idx is None:
            rays_idx = self._picks.shared_pixels(torch.randperm(self.views.H * self.views.W, device=self.poses.device)[: self.rand_rays // V])
        self._idx.copy_(rays_idx)
        self._view.fill_(self._picks.view(V) if view is None else int(view))
        return self.stage.step()

    def run(self, n_iters=None, picks=None):
        logs = {k: [] for k in self._keys}
        for it in range(self.max_iter if n_iters is None else int(n_iters)):
            ret = self.step(*(picks[it] if picks is not None else (None, None)))
            for k in self._keys:
                logs[k].append(ret[k].detach().reshape(()).clone())
        return {("all" if k == "loss_all" else k): torch.stack(v) for k, v in logs.items()}


class InitLoop:
    """`Initializer.run` (pipelines/Initialization.py:139-226) for the two initial views with given poses (the essential-matrix
    pose initialisation of `Initializer.__init__` is pycolmap's: caller side): per iteration the matched key points of each
    view are sphere-traced onto the surface and projected into the OTHER view (`Camera.proj_cam_i`, Camera.py:168-178) --
    reproj_error = mean ||uv_proj - key point||, sdf_surf = mean |SDF at the end of the tracks|, both over the 2 n tracks; the
    gradient reaches the SDF field through the traced depth -- then the render with the cameras' poses, eikonal over EVERY
    normal, rgb, depth consistency; one backward; Adam over the two fields (the poses are not in this optimizer,
    Initialization.py:124-125); ExponentialLR.  `triangulate()` is the block after the loop (Initialization.py:182-213).

    views: a two-view `TrackedViews` whose key points are the inlier matches, row j of view 0 <-> row j of view 1 (what
    `kypts[mch_msks[..., 0]][inlier_msks]` and `cam_i.kypts[mch_msks[..., 1]][inlier_msks]` select); track_ids / xyzs unused."""

    def __init__(self, opt, renderer, sdf_field, rad_field, views, weights, lr_sdf, lr_sdf_end, lr_color, max_iter, rand_rays,
                 capture=False, static_trips=None, sdf_filter=True, device_picks=False):
        get = (lambda k: weights.get(k)) if isinstance(weights, dict) else (lambda k: getattr(weights, k, None))
        if len(views.keypoints) != 2 or views.keypoints[0].shape != views.keypoints[1].shape:
            raise ValueError("ls2fm.stage.InitLoop: two views with the same number of matched key points")
        self.sdf, self.views, self.max_iter, self.rand_rays = sdf_field, views, int(max_iter), int(rand_rays)
        self.sdf_filter = bool(sdf_filter)
        self.poses = (views.poses if views.poses.shape[-1] == 4 else _cam.lie.se3_to_SE3(views.poses)).detach()
        from . import fused as _fused
        self.static = _fused.available(sdf_field, views.images) if static_trips is None else static_trips
        self.w_reproj = 0.0 if get("reproj_error") is None else 10.0 ** float(get("reproj_error"))
        self.w_surf = 0.0 if get("sdf_surf") is None else 10.0 ** float(get("sdf_surf"))
        with torch.no_grad():              # the poses do not move during the loop: the key points' rays are formed once
            self._kp_rays = [keypoint_rays(self.poses[v], views.intrinsic, views.keypoints[v]) for v in range(2)]
        self.stage = RenderStage(opt, renderer, sdf_field, rad_field, weights=weights, lr=lr_sdf, lr_end=lr_sdf_end, max_iter=max_iter,
                                 lr_color=lr_color, capture=capture, extra_loss=self._extra, eikonal_over="all", extra_prepare=self._prepare,
                                 input_fn=lambda: self._fixed.pick(self._idx), share_gradients=bool(self.static))
        self._picks = _HostPicks(views.H * views.W, self.rand_rays // 2, self.poses.device, tail=0)
        self._idx = self._picks.dev
        self.device_picks = bool(device_picks)
        self._fixed = _FixedPoseRays(views, self.poses)
        self._keys = ("loss_all", "PSNR", "rgb_loss", "DC_loss", "eikonal_loss", "sdf_surf", "reproj_error")
        self._surface = self._finish = None
        self._early = [_EarlyTrace(sdf_field, slot=1), _EarlyTrace(sdf_field, slot=2)]
        self._match = None
        if self.poses.is_cuda and self.poses.dtype == torch.float32:
            # the fixed operands of the fused match node: segment v = the key points of view v, seen through the OTHER view
            with torch.no_grad():
                n = views.keypoints[0].shape[0]
                cen = torch.cat([self._kp_rays[v][0].reshape(-1, 3) for v in range(2)]).float().contiguous()
                ray = torch.cat([self._kp_rays[v][1].reshape(-1, 3) for v in range(2)]).float().contiguous()
                uv = torch.cat([views.keypoints[1 - v].reshape(-1, 2) for v in range(2)]).float().contiguous()
                pose_o = torch.stack([self.poses[1 - v] for v in range(2)]).float().contiguous()
            self._match = (cen, ray, uv, pose_o, host_intrinsic(views.intrinsic), n)
            self._surface_buf = torch.zeros(2, n, 3, device=self.poses.device)

    def _prepare(self):
        if self.static:
            for v in range(2):
                self._early[v].launch(*self._kp_rays[v])

    def _extra(self, ret):
        """the explicit-match terms (Initialization.py:154-160, 252-255), already weighted"""
        if self._match is not None:
            # one node each way for the whole block below (ls2fm_match_term_fwd / _bwd): ~70 graph nodes of a captured iteration
            ds, sls, fins = [], [], []
            for v in range(2):
                center, ray = self._kp_rays[v]
                early = self._early[v].take()
                d, sdf_last, _, fin = early if early is not None else self.sdf.sphere_tracing(center, ray, self.sdf, static_trips=self.static)
                ds.append(d); sls.append(sdf_last); fins.append(fin.reshape(-1).bool())
            terms = _MatchTerm.apply(self._match, self._surface_buf.view(-1, 3), *ds, *sls)
            ret["reproj_error"], ret["sdf_surf"] = terms[0], terms[1]
            self._surface, self._finish = self._surface_buf, torch.stack(fins)
            return _weighted_pair(ret["reproj_error"], ret["sdf_surf"], self.w_reproj, self.w_surf)
        errs, sdfs, surface, finish = [], [], [], []
        for v in range(2):
            center, ray = self._kp_rays[v]
            early = self._early[v].take()
            d, sdf_last, _, fin = early if early is not None else self.sdf.sphere_tracing(center, ray, self.sdf, static_trips=self.static)
            pts = center + ray * d.reshape(1, -1, 1)                                             # Camera.py:136
            o = 1 - v
            uv = _cam.cam2img(_cam.world2cam(pts, self.poses[o:o + 1]), self.views.intrinsic.unsqueeze(0))
            uv = (uv / (uv[..., 2:] + 1e-6))[..., :2]                                            # Camera.py:173
            errs.append((uv[0] - self.views.keypoints[o]).norm(dim=-1))
            sdfs.append(sdf_last.reshape(-1))
            surface.append(pts[0].detach())
            finish.append(fin.reshape(-1).bool())
        ret["reproj_error"] = torch.cat(errs).mean()
        ret["sdf_surf"] = torch.cat(sdfs).abs().mean()
        self._surface, self._finish = torch.stack(surface), torch.stack(finish)
        return self.w_reproj * ret["reproj_error"] + self.w_surf * ret["sdf_surf"]

    def step(self, rays_idx=None):
        if rays_idx is None and not self.device_picks:
            self._picks.draw()
            rays_idx = self._idx
        elif rays_idx is None:
            rays_idx = torch.randperm(self.views.H * self.views.W, device=self.poses.device)[: self.rand_rays // 2]
        if rays_idx is not self._idx:
            self._idx.copy_(rays_idx)
        return self.stage.step()

    def run(self, n_iters=None, picks=None):
        logs = {k: [] for k in self._keys}
        for it in range(self.max_iter if n_iters is None else int(n_iters)):
            ret = self.step(picks[it] if picks is not None else None)
            for k in self._keys:
                logs[k].append(ret[k].detach().reshape(()).clone())
        return {("all" if k == "loss_all" else k): torch.stack(v) for k, v in logs.items()}

    @torch.no_grad()
    def triangulate(self):
        """-> (points [n,3], kept [n] bool): the mean of the two views' traced surface points of the LAST iteration; kept =
        within mean + 3 sigma of the two-view distance and (sdf_filter) finished in at least one view (Initialization.py:183-191).
        The caller adds `points[kept]` to its point set with the feature tracks (view 0 key point j, view 1 key point j)."""
        if self._surface is None:
            raise RuntimeError("ls2fm.stage.InitLoop.triangulate: run at least one step first")
        diff = (self._surface[0] - self._surface[1]).norm(dim=-1)
        kept = diff < diff.mean() + 3 * diff.std()
        if self.sdf_filter:
            kept = kept & (self._finish[0] | self._finish[1])
        return (self._surface[0] + self._surface[1]) / 2, kept


class GeoInitLoop:
    """`Registration.geo_init_nf` (pipelines/Registration.py:133-296): a NEW view against registered ones, SDF field only, no
    render.  Per iteration ONE sphere tracing over the matched key points of every (new, registered) pair from both sides; for
    the matches without a 3-D point the two traced points are re-projected into the other view -- outliers by the finish flags
    and the 1x / 2x / 4x `reproj_max` bounds, as device-side masks (the reference's `(~mask).sum() > 0` is a select here) --,
    for the matches that have one the traced point is compared with it; sdf_surf over the tracks' last values and the existing
    points near the surface; eikonal over the existing points, the track points and the random along-ray points of
    `sphere_tracing` (fixed shape + mask, ls2fm.models.SDF); one backward; Adam over the SDF field; ExponentialLR with the
    reference's factor (lr_end / lr) ** (1 / max_iter) over its 5 * max_iter iterations.  `triangulate()` is the block after
    the loop.

    pairs: per registered view a dict  view (index into poses), kp_new [n,2], kp_src [n,2] (row j <-> row j: the inlier
    matches), point_id [n] long (index into xyzs of the 3-D point the new view's key point already has, -1: none)."""

    def __init__(self, opt, sdf_field, poses, intrinsic, new_view, pairs, xyzs, weights, lr_sdf, lr_sdf_end, max_iter, reproj_max=15.0,
                 capture=False):
        """capture=True: the first `step` records the whole iteration -- tracing, the fixed-shape sample points and their mask,
        three point-query nodes, every term, the backward, Adam with the schedule on the device -- into ONE hipGraph; later
        steps write the iteration's uniform draws into a persistent buffer and replay it."""
        get = (lambda k: weights.get(k)) if isinstance(weights, dict) else (lambda k: getattr(weights, k, None))
        w = lambda k: 0.0 if get(k) is None else 10.0 ** float(get(k))
        self.w_reproj, self.w_tracing, self.w_surf, self.w_eik = w("reproj_error"), w("tracing_loss"), w("sdf_surf"), w("eikonal_loss")
        self.sdf, self.intrinsic, self.xyzs = sdf_field, intrinsic, xyzs
        self.poses = (poses if poses.shape[-1] == 4 else _cam.lie.se3_to_SE3(poses)).detach()
        self.new_view, self.reproj_max = int(new_view), float(reproj_max)
        self.n_iters = 5 * int(max_iter)                                          # Registration.py:140
        self.pairs = []
        centers, rays = [[], []], [[], []]
        at = 0
        with torch.no_grad():
            for pr in pairs:
                n = pr["kp_new"].shape[0]
                c0, r0 = keypoint_rays(self.poses[self.new_view], intrinsic, pr["kp_new"])        # the new view's side (ret0)
                c1, r1 = keypoint_rays(self.poses[int(pr["view"])], intrinsic, pr["kp_src"])     # the registered view's side (ret1)
                centers[0].append(c0); rays[0].append(r0); centers[1].append(c1); rays[1].append(r1)
                pid = pr["point_id"].long()
                self.pairs.append(dict(view=int(pr["view"]), lo=at, hi=at + n, kp_new=pr["kp_new"], kp_src=pr["kp_src"], has=pid >= 0,
                                       target=xyzs[pid.clamp_min(0)], n_has=int((pid >= 0).sum())))
                at += n
            self.center = torch.cat([torch.cat(centers[0], dim=1), torch.cat(centers[1], dim=1)], dim=0)     # [2, sum n, 3]
            self.ray = torch.cat([torch.cat(rays[0], dim=1), torch.cat(rays[1], dim=1)], dim=0)
        self.n_frames = sum(1 for p in self.pairs if p["n_has"] > 0)
        self.params = [p for p in sdf_field.parameters() if p.requires_grad]
        self.optim = FusedAdam([dict(params=self.params, lr=lr_sdf)], lr=lr_sdf,
                               scheduled_gamma=(lr_sdf_end / lr_sdf) ** (1.0 / int(max_iter)))       # Registration.py:33
        self._one = torch.ones((), device=xyzs.device)
        self._last = None
        self._keys = ("loss_all", "reproj_error", "tracing_loss", "sdf_surf", "eikonal_loss")
        self.capture = bool(capture)
        self._graph = None
        self._u = torch.zeros(self.center.shape[0] * self.center.shape[1], device=xyzs.device)      # the draws of SDF.py:217

    def step(self, sample_u=None):
        """one iteration; sample_u [2 * n]: the uniform draws that place the along-ray eikonal points (parity tests replay the
        reference's); default: fresh device-side draws"""
        if not self.capture:
            ret, self._last = self._iteration(sample_u)
            return ret
        if sample_u is None:
            self._u.uniform_()
        else:
            self._u.copy_(sample_u.reshape(-1))
        if self._graph is None:
            from .graph import CapturedStep
            snap = _snapshot_training_state(self.params, self.optim)      # the capture warms up by stepping for real
            self._graph = CapturedStep(lambda: self._iteration(self._u), params=self.params)
            _restore_training_state(self.params, self.optim, snap)
        ret, self._last = self._graph.replay()
        self.optim.replayed(1)
        return ret

    def _project(self, pts, view):
        uv = _cam.cam2img(_cam.world2cam(pts.unsqueeze(0), self.poses[view:view + 1]), self.intrinsic.unsqueeze(0))[0]
        return (uv / (uv[..., 2:] + 1e-6))[..., :2]

    def _iteration(self, sample_u=None):
        for p in self.params:
            p.grad = None
        sdf = self.sdf
        d, sdf_last, sampled, fin = sdf.sphere_tracing(self.center, self.ray, sdf, static_trips=True, want_samples=True, sample_u=sample_u)
        pts = self.center + self.ray * d.reshape(2, -1, 1)                                         # Registration.py:190
        fin = fin.reshape(2, -1).bool()
        zero = torch.zeros((), device=pts.device)
        reproj, n_reproj, tracing, last = zero, zero, zero, []
        rmax = self.reproj_max
        for pr in self.pairs:
            p0, p1 = pts[0, pr["lo"]:pr["hi"]], pts[1, pr["lo"]:pr["hi"]]
            f0, f1 = fin[0, pr["lo"]:pr["hi"]], fin[1, pr["lo"]:pr["hi"]]
            e0 = (self._project(p0, pr["view"]) - pr["kp_src"]).norm(dim=-1)                      # new view's point in the registered view
            e1 = (self._project(p1, self.new_view) - pr["kp_new"]).norm(dim=-1)
            bad = ((f0 & (e0 > rmax)) & (f1 & (e1 > rmax))) | ((e0 > 2 * rmax) & (e1 > 2 * rmax)) | (e0 > 4 * rmax) | (e1 > 4 * rmax)
            keep = (~pr["has"]) & ~bad
            cnt = keep.sum()
            some = cnt > 0
            pair_err = ((e0 * keep).sum() + (e1 * keep).sum()) / (2 * cnt.clamp_min(1))
            reproj = reproj + torch.where(some, pair_err, zero)
            n_reproj = n_reproj + some
            dist = (pr["target"] - p0).norm(dim=-1)
            if pr["n_has"] > 0:
                tracing = tracing + (dist * pr["has"]).sum() / pr["n_has"]
            last.append((p0.detach(), p1.detach(), f0, f1, keep, dist.detach()))
        # the existing points: near-surface ones enter sdf_surf, all of them the eikonal term (Registration.py:257-262)
        sdf_exist = sdf.infer_sdf(self.xyzs).reshape(-1)
        near = sdf_exist.detach().abs() < float(sdf.sdf_threshold)
        grad_exist = sdf.gradient(self.xyzs).norm(dim=-1).reshape(-1)
        sdf_surf = ((sdf_exist.abs() * near).sum() + sdf_last.reshape(-1).abs().sum()) / (near.sum() + sdf_last.numel())
        live = sdf.last_sample_mask.reshape(-1)
        grad_sampled = sdf.gradient(sampled.reshape(-1, 3)).norm(dim=-1).reshape(-1)
        eik = ((grad_exist - 1).abs().sum() + ((grad_sampled - 1).abs() * live).sum()) / (grad_exist.numel() + live.sum())
        any_reproj = n_reproj > 0
        ret = dict(reproj_error=torch.where(any_reproj, reproj / n_reproj.clamp_min(1), zero),
                   tracing_loss=tracing / max(self.n_frames, 1), sdf_surf=sdf_surf, eikonal_loss=eik)
        ret["loss_all"] = (self.w_reproj * ret["reproj_error"] + self.w_tracing * ret["tracing_loss"] + self.w_surf * sdf_surf
                           + self.w_eik * eik)
        ret["loss_all"].backward(gradient=self._one)
        self.optim.step()
        return ret, last

    def run(self, n_iters=None, draws=None):
        logs = {k: [] for k in self._keys}
        for it in range(self.n_iters if n_iters is None else int(n_iters)):
            ret = self.step(draws[it] if draws is not None else None)
            for k in self._keys:
                logs[k].append(ret[k].detach().reshape(()).clone())
        return {("all" if k == "loss_all" else k): torch.stack(v) for k, v in logs.items()}

    @torch.no_grad()
    def triangulate(self):
        """-> per pair (points [n,3], kept [n] bool): the block after the loop (Registration.py:271-294) on the LAST iteration's
        traced points -- a new match becomes a 3-D point (the mean of its two traced points) when the two agree within mean + std of
        the existing matches' tracing distances, or both tracks finished"""
        if self._last is None:
            raise RuntimeError("ls2fm.stage.GeoInitLoop.triangulate: run at least one step first")
        record = torch.cat([dist[pr["has"]] for pr, (_, _, _, _, _, dist) in zip(self.pairs, self._last)])
        thr = record.mean() + record.std()
        out = []
        for p0, p1, f0, f1, keep, _ in self._last:
            diff = (p0 - p1).norm(dim=-1)
            out.append(((p0 + p1) / 2, keep & ((diff <= thr) | (f0 & f1))))
        return out


class _Reproject(torch.autograd.Function):
    """ls2fm_reproject_fwd / _bwd: the re-projection term of a BA iteration (BA.py:126-147, 199-202) as one node
    -> (reproj scalar, counted bool [n])"""

    @staticmethod
    def forward(ctx, points, poses, view_start, k_host, obs_uv, sdf, bound):
        import ctypes
        from . import _lib
        lib = _lib.load()
        x, ps = points.detach().float().contiguous(), poses.detach().float().contiguous()
        _lib.require_device(x)
        n, n_views, dev = x.shape[0], ps.shape[0], x.device
        err = torch.empty(n, device=dev)
        on = torch.empty(n, device=dev, dtype=torch.uint8)
        sums = torch.empty(4, device=dev, dtype=torch.float64)
        ws = torch.empty(int(lib.ls2fm_reproject_workspace_bytes(n_views)) // 8, device=dev, dtype=torch.float64)
        sdf_c = None if sdf is None else sdf.detach().float().contiguous().reshape(-1)
        _lib.check(lib.ls2fm_reproject_fwd(_lib.ptr(x), _lib.ptr(ps), _lib.ptr(view_start), n_views, k_host, _lib.ptr(obs_uv), _lib.ptr(sdf_c),
                                           float(bound), n, _lib.ptr(err), _lib.ptr(on), _lib.ptr(sums), _lib.ptr(ws),
                                           _lib.stream_ptr()), "ls2fm_reproject_fwd")
        ctx.save_for_backward(x, ps, view_start, obs_uv, sdf_c if sdf_c is not None else x.new_empty(0), sums)
        ctx.k_host, ctx.bound, ctx.has_sdf = k_host, float(bound), sdf_c is not None
        counted = on.view(torch.bool)
        ctx.mark_non_differentiable(counted)
        ctx.set_materialize_grads(False)         # (no zero fill for the bool output's "gradient" in the backward)
        return sums[3].float(), counted

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_reproj, _):
        from . import _lib
        lib = _lib.load()
        x, ps, view_start, obs_uv, sdf_c, sums = ctx.saved_tensors
        if d_reproj is None:
            return (None,) * 7
        d_x, d_ps = torch.empty_like(x), torch.empty_like(ps)
        g = d_reproj.detach().float().reshape(1).contiguous()
        _lib.check(lib.ls2fm_reproject_bwd(_lib.ptr(x), _lib.ptr(ps), _lib.ptr(view_start), ps.shape[0], ctx.k_host, _lib.ptr(obs_uv),
                                           _lib.ptr(sdf_c) if ctx.has_sdf else None, ctx.bound, x.shape[0], _lib.ptr(sums), _lib.ptr(g),
                                           _lib.ptr(d_x), _lib.ptr(d_ps), _lib.stream_ptr()), "ls2fm_reproject_bwd")
        return d_x, d_ps, None, None, None, None, None


def reprojection_term(points, poses, view_start, intrinsic_host, obs_uv, sdf=None, bound=float("inf")):
    """points [n,3] (observations sorted by view: view v owns rows view_start[v] : view_start[v + 1], int32 [V + 1] on the device),
    poses [V,3,4] world-to-camera, intrinsic_host: the 3 x 3 K as 9 host floats (`ctypes` array, see `host_intrinsic`), obs_uv
    [n,2] -> (reproj, counted [n] bool): 0.5 mean(2 log(1 + err^2 / 4)) + 0.5 mean(err) over the observations with
    |sdf| < bound and a finite projection (BA.py:126-147, 199-202); differentiable w.r.t. points and poses"""
    return _Reproject.apply(points, poses, view_start, intrinsic_host, obs_uv.float().contiguous(), sdf, bound)


def host_intrinsic(intrinsic):
    import ctypes
    return (ctypes.c_float * 9)(*[float(v) for v in intrinsic.detach().reshape(-1).cpu().tolist()])


class BALoop:
    """`BA` in mode "sfm_refine" with several cameras (pipelines/BA.py:24-218; optim_split: rotation / translation parameters
    with their own rates, BA.py:66-75): per iteration the POINT side -- tracked points projected onto the surface
    (get_surface_pts), re-evaluated (infer_sdf), re-projected through the live poses against their key points (robust mean over
    |sdf| < 2 thr), the adaptive weight 10^1 when the error exceeds 10 px (BA.py:163-166) -- then the render side with the
    poses detached (BA.py:150-151), eikonal over mask_bg, sdf_surf over the points, tracing consistency; one backward; Adam over
    {rotations, translations, SDF field, radiance field}; ExponentialLR; the points are replaced by their projections
    (BA.py:181)."""

    def __init__(self, opt, renderer, sdf_field, rad_field, views, weights, lr_sdf, lr_sdf_end, lr_color, lr_pose_r, lr_pose_t,
                 max_iter, rand_rays, capture=False, static_trips=None, distributed=False, device_picks=False):
        """distributed=True: BASELINE.json configs[3] -- the render rays of the registered views sharded by view over the ranks of
        the process group, one gradient all-reduce per iteration (fields' flat buffer + the pose groups); the point side,
        re-projection and tracing consistency are evaluated on every rank (same inputs) with weight 1 / world in the backward"""
        get = (lambda k: weights.get(k)) if isinstance(weights, dict) else (lambda k: getattr(weights, k, None))
        self.opt, self.sdf, self.views = opt, sdf_field, views
        self.max_iter, self.rand_rays = int(max_iter), int(rand_rays)
        se3 = views.poses if views.poses.shape[-1] == 6 else _cam.lie.SE3_to_se3(views.poses)
        self.rot = torch.nn.Parameter(se3[:, :3].detach().clone())
        self.trans = torch.nn.Parameter(se3[:, 3:].detach().clone())
        # the observation list of util.get_idx3d_camset (utils/util.py:450-464): per view, every key point with a 3-D point
        self.obs_view = torch.cat([torch.full((t.shape[0],), v, dtype=torch.long) for v, t in enumerate(views.track_ids)]).to(se3.device)
        self.obs_point = torch.cat(views.track_ids).to(se3.device)
        self.obs_uv = torch.cat(views.keypoints).to(se3.device)
        counts = [0] + [t.shape[0] for t in views.track_ids]
        self.view_start = torch.tensor(counts, dtype=torch.int32).cumsum(0).to(torch.int32).to(se3.device)       # observations are sorted by view
        self._k_host = host_intrinsic(views.intrinsic)
        # the loop's OWN copy of the points (BA.py:77): it is what gets projected and replaced every iteration, while the
        # tracing consistency keeps comparing against the point set's coordinates, which do not move during the loop (and not
        # after it either: Point3DSet.update_xyzs never runs its lazy map, SURVEY C-13)
        self.xyzs_all = views.xyzs.detach().clone()
        self.w_surf = 0.0 if get("sdf_surf") is None else 10.0 ** float(get("sdf_surf"))
        self.w_reproj_lo = 0.0 if get("reproj_error") is None else 10.0 ** float(get("reproj_error"))
        self.sdf_threshold = float((float(opt.data.bound_max[0]) - float(opt.data.bound_min[0])) / 10 / int(opt.Res))
        from . import fused as _fused
        static = _fused.available(sdf_field, views.images) if static_trips is None else static_trips
        self.tracing = TracingConsistency(sdf_field, views, get("tracing_loss"), None, static_trips=static)
        self.tracing.use_sdfs = False
        self.stage = RenderStage(opt, renderer, sdf_field, rad_field, weights=weights, lr=lr_sdf, lr_end=lr_sdf_end, max_iter=max_iter,
                                 lr_color=lr_color, capture=capture, extra_loss=self._extra, eikonal_over="bg",
                                 extra_params=[dict(params=[self.rot], lr=lr_pose_r), dict(params=[self.trans], lr=lr_pose_t)],
                                 extra_prepare=self._prepare, input_fn=self._inputs, share_gradients=bool(static),
                                 shard_views=bool(distributed), reducer="auto" if distributed else None)
        k = self.rand_rays // se3.shape[0]
        self._picks = _HostPicks(views.H * views.W, k, se3.device, tail=1, shared=distributed)
        self._idx, self._view = self._picks.dev[:k], self._picks.dev[k:]
        self.device_picks = bool(device_picks)
        self._keys = ("loss_all", "PSNR", "rgb_loss", "DC_loss", "eikonal_loss", "sdf_surf", "tracing_loss", "reproj_error", "w_reproj")
        self._render_poses = torch.zeros(se3.shape[0], 3, 4, device=se3.device)
        # the key-point rays of the tracing consistency come from the CAMERAS' own poses (Camera.get_pts3D -> get_pose,
        # Camera.py:131), which the loop only writes back after its last iteration (BA.py:184-185): fixed during the loop
        self._camera_poses = _cam.lie.se3_to_SE3(se3).detach()

    def _inputs(self):
        self.tracing.select(self._view, self._camera_poses)
        with torch.no_grad():            # the render poses are detached (BA.py:150-151): exponential + ray pick in one launch
            return _pick_rays(self.views, None, self._idx, se3=torch.cat([self.rot, self.trans], dim=1), poses_out=self._render_poses)

    def _prepare(self):
        """ahead of the render, as in BA.run_ba (BA.py:117-131 come before its render call): the tracing consistency's key-point
        tracing starts on a stream of its own, and the point side's two field queries are issued -- their autograd nodes then
        PRECEDE the render's, so that in the backward they run after it and add into its gradient buffer (ls2fm.fused: one
        gradient buffer per backward pass) instead of producing dense gradients of their own"""
        self.tracing.prepare()
        xyzs_new, _ = self.sdf.get_surface_pts(self.xyzs_all[self.obs_point])
        self._point_side = (xyzs_new, self.sdf.infer_sdf(xyzs_new, mode="ret_sdf").view(-1, 1))

    def _extra(self, ret):
        """the terms BA.run_ba forms outside the render (BA.py:119-147) + the tracing consistency, already weighted"""
        if getattr(self, "_point_side", None) is not None:
            (xyzs_new, sdfs), self._point_side = self._point_side, None
        else:
            xyzs_new, _ = self.sdf.get_surface_pts(self.xyzs_all[self.obs_point])
            sdfs = self.sdf.infer_sdf(xyzs_new, mode="ret_sdf").view(-1, 1)
        poses = _cam.se3_to_SE3_fused(torch.cat([self.rot, self.trans], dim=1))                  # [V,3,4]: the live poses
        if xyzs_new.is_cuda:
            # one fused node each way for the per-observation block (projection, pixel error, on-surface / finite mask, robust mean)
            reproj, _ = reprojection_term(xyzs_new, poses, self.view_start, self._k_host, self.obs_uv, sdfs, 2 * self.sdf_threshold)
        else:
            reproj = self._reproj_torch(xyzs_new, poses, sdfs)
        if xyzs_new.is_cuda and sdfs.dtype == torch.float32:
            # BA.py:160-170 as one node each way: sdf_surf, the adaptive weight, the weighted sum with the tracing loss
            on = 1.0 if self.w_reproj_lo else 0.0
            tl = self.tracing(ret, raw=True)           # unweighted on its fused path (then it IS ret["tracing_loss"]), else weighted
            raw = tl is ret.get("tracing_loss")
            surf, w_reproj, extra = _BATerms.apply(reproj, sdfs, tl if raw else None, 10.0, on, 10.0 * on, self.w_surf,
                                                   self.tracing.w_tracing if raw else 0.0)
            ret["reproj_error"], ret["w_reproj"], ret["sdf_surf"] = reproj, w_reproj, surf
            self._new_points = xyzs_new.detach()
            return extra if raw else extra + tl
        if getattr(self, "_w_hi", None) is None or self._w_hi.device != reproj.device:       # constants once, not two fill kernels per iteration
            on = 1.0 if self.w_reproj_lo else 0.0
            self._w_hi, self._w_lo = torch.full((), 10.0 * on, device=reproj.device), torch.full((), on, device=reproj.device)
        w_reproj = torch.where(reproj.detach() > 10, self._w_hi, self._w_lo)                  # BA.py:163-166
        ret["reproj_error"], ret["w_reproj"] = reproj, w_reproj
        ret["sdf_surf"] = sdfs.abs().mean()
        self._new_points = xyzs_new.detach()
        return w_reproj * reproj + self.w_surf * ret["sdf_surf"] + self.tracing(ret)

    def _reproj_torch(self, xyzs_new, poses, sdfs):
        """the same term written with torch ops (the reference's lines: BA.py:126-147, 199-202)"""
        in_cam = _cam.world2cam(xyzs_new.unsqueeze(1), poses[self.obs_view])
        uv = _cam.cam2img(in_cam, self.views.intrinsic.expand(in_cam.shape[0], 3, 3))
        uv = (uv / (uv[..., 2:] + 1e-6))[..., :2].squeeze(1)
        on_surface = (sdfs.abs() < 2 * self.sdf_threshold).squeeze(-1) & ~torch.isinf(uv).any(dim=-1)
        err = torch.where(on_surface, (uv - self.obs_uv).norm(dim=-1), torch.zeros((), device=uv.device))
        n = on_surface.sum()
        robust = torch.where(on_surface, 2 * torch.log(1 + err ** 2 / 4), torch.zeros((), device=uv.device))
        return torch.where(n > 0, 0.5 * robust.sum() / n.clamp_min(1) + 0.5 * err.sum() / n.clamp_min(1), torch.zeros((), device=uv.device))

    def step(self, rays_idx=None, view=None):
        V = self.rot.shape[0]
        if rays_idx is None and view is None and not self.device_picks:
            self._picks.draw(self._picks.view(V))
        else:
            if rays_idx is None:
                rays_idx = self._picks.shared_pixels(torch.randperm(self.views.H * self.views.W, device=self.rot.device)[: self.rand_rays // V])
            self._idx.copy_(rays_idx)
            self._view.fill_(self._picks.view(V) if view is None else int(view))
        ret = self.stage.step()
        with torch.no_grad():
            self.xyzs_all[self.obs_point] = self._new_points                                     # BA.py:181
        return ret

    def run(self, n_iters=None, picks=None):
        logs = {k: [] for k in self._keys}
        for it in range(self.max_iter if n_iters is None else int(n_iters)):
            ret = self.step(*(picks[it] if picks is not None else (None, None)))
            for k in self._keys:
                logs[k].append(ret[k].detach().reshape(()).clone())
        return {("all" if k == "loss_all" else k): torch.stack(v) for k, v in logs.items()}

    def poses_se3(self):
        return torch.cat([self.rot, self.trans], dim=1).detach()
