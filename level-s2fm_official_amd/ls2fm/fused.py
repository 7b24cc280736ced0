"""Fused-kernel front end: one autograd node for the whole of Renderer.forward, the no-grad SDF
evaluation and the sphere-tracing root-find loop.  The kernels read the nn.Parameters of the reference-shaped
modules directly (weight_v / weight_g / bias, the flat hash tables, beta) and return gradients in that
very parametrisation, so nothing but this one node sits between the optimizer and the HIP code.

Gating: the fused kernels implement the reference's network sizes (hidden 64, 16 features, 2 features per
level, <= 16 levels, affine radiance decoder) with uniform sampling and no background-sphere `min`, including the
gradients w.r.t. the camera rays.  Anything else is served by the general autograd
composition (HIP hash-grid op + torch dense layers), never by a CPU path.
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

_DISABLE = os.environ.get("LS2FM_DISABLE_FUSED", "0") == "1"
_POISON = os.environ.get("LS2FM_POISON_WS", "0") == "1"
# Dual field: the render gathers from an ENTRY-INTERLEAVED copy of the two hash tables ([entry][sdf f0 f1 | rad f0 f1]): one
# 16-byte gather per corner serves both grids -- the gather pass is bound by the L2 -> L1 line rate, not by bytes, and this halves
# its requests.  The two nn.Parameters stay what the reference has (flat fp32 vectors, state_dict keys of SURVEY App. E); the
# copy is a derived buffer that is kept current in one of two ways:
#   * `ls2fm.optim.FusedAdam` writes every updated table value into it in its own pass over the tables (the Parameters carry a
#     `_ls2fm_mirror` record; ls2fm_adam_step_mirrored) -- a training loop pays no rebuild;
#   * anything else that changes a table (another optimizer, load_state_dict, an in-place op: all bump Tensor._version)
#     makes the next render rebuild it (ls2fm_interleave_tables, ~45 us).
# LS2FM_DUAL_TABLE: "version" (default) as above; "always": rebuilt on every forward (for code that writes the tables behind
# autograd's back, e.g. through `.data`); "off": gather from the two tables (8 bytes twice per corner).
_DUAL_TABLE = os.environ.get("LS2FM_DUAL_TABLE", "version")
if _DUAL_TABLE not in ("version", "always", "off"):
    raise RuntimeError(f"LS2FM_DUAL_TABLE={_DUAL_TABLE!r}: expected version, always or off")


class TableMirror:
    """the interleaved copy of one (SDF table, second table) pair and the versions of the two Parameters it reflects"""
    __slots__ = ("table", "versions", "ptrs", "by_optimizer")

    def __init__(self):
        self.table, self.versions, self.ptrs, self.by_optimizer = None, [None, None], [None, None], False

    def fresh(self, sdf_table, rad_table) -> bool:
        return (self.table is not None and self.ptrs == [sdf_table.data_ptr(), rad_table.data_ptr()]
                and self.versions == [sdf_table._version, rad_table._version])

    def written_by_optimizer(self, slot, p):
        """FusedAdam just wrote `p`'s updated values into the copy (and bumped p's version)"""
        self.versions[slot] = p._version
        self.by_optimizer = True


# ------------------------------------------------------------------------------------------------ gating
def _geometry_ok(mlp, in_dim) -> bool:
    layers = getattr(mlp, "mlp", None)
    if layers is None or len(layers) != 2 or getattr(mlp, "skip", []):
        return False
    l0, l1 = layers
    return (hasattr(l0, "weight_g") and hasattr(l1, "weight_g")
            and tuple(l0.weight_v.shape) == (_lib.HIDDEN, in_dim)
            and tuple(l1.weight_v.shape) == (_lib.FEAT + 1, _lib.HIDDEN))


def supported(opt, sdf_field, rad_field=None) -> bool:
    """True when the fused kernels implement this configuration."""
    if _DISABLE:
        return False
    enc = sdf_field.embed_fn.embedder_obj
    desc = enc.desc
    if desc.n_features != 2 or desc.n_levels > _lib.MAX_LEVELS:
        return False
    if not _geometry_ok(sdf_field.SDF_MLP, 3 + 2 * desc.n_levels):
        return False
    if rad_field is None:
        return True              # sdf_eval / sphere_trace (they implement the background-sphere min)
    if opt.data.inside == True and opt.data.bg_sdf == True:  # noqa: E712
        return False             # fused render: no background-sphere min
    if opt.SDF.VolSDF.volsdf_sampling != False or not 1 <= int(opt.SDF.VolSDF.sample_intvs) <= 512:  # noqa: E712
        return False
    dual = opt.Ablate_config.dual_field == True  # noqa: E712
    # the table-gradient scatter sorts into at most 128 slabs per level (csrc/bin_scatter.hip: 8192-entry slabs, 4096 when
    # both grids share a pass): tables up to 2^20 (2^19 dual) entries per level; larger ones take the composed form
    if max(desc.size[:desc.n_levels]) > (128 << (12 if dual else 13)):
        return False
    if dual:
        d2 = rad_field.embed_fn.embedder_obj.desc
        if d2.n_levels != desc.n_levels or not _geometry_ok(rad_field.Geo_enc, 3 + 2 * d2.n_levels):
            return False
    dec = rad_field.Rad_dec
    rad_in = 33 + _lib.FEAT * (2 if dual else 1)
    shapes = [tuple(layer.weight_v.shape) if hasattr(layer, "weight_v") else None for layer in dec.mlp_radiance]
    return shapes == [(64, rad_in), (64, 64), (3, 64)] and not dec.hidden_relu


def available(field, probe) -> bool:
    return (not _DISABLE) and probe.is_cuda and supported(field.opt, field) and _HAS_SDF_EVAL


def can_eval_without_graph(sdf_field, xyz) -> bool:
    if not xyz.is_cuda or not _HAS_SDF_EVAL or not supported(sdf_field.opt, sdf_field):
        return False
    if not torch.is_grad_enabled():
        return True
    return not xyz.requires_grad and not any(p.requires_grad for p in sdf_field.parameters())


class _Plan:
    """Everything about one (renderer, SDF field, radiance field, opt) combination that does not change from call to
    call: the gating verdict, the descriptor structs, the parameter list.  Rebuilding these costs ~0.2 ms of Python
    per step -- as much as the device work of a kernel -- so they are cached.  A plan lives in a dict ON the renderer
    object (it dies with it), holds the two field modules weakly (it is dropped when either is freed, so a recycled
    id() can never resurrect it) and is only reused while (a) every option / descriptor value the kernels read and (b)
    the identity of every Parameter and grid descriptor are unchanged."""
    __slots__ = ("key", "ok", "cfg", "ts", "ws_bytes", "pkey", "pstruct", "dual_table", "dual_key", "sdf_ref", "rad_ref",
                 "__weakref__")


def _plan_key(renderer, opt, sdf_field, rad_field):
    """every value read from `opt` / the modules that ends up in a descriptor struct or in the gating verdict"""
    data, vol = opt.data, opt.SDF.VolSDF
    return (int(vol.sample_intvs), opt.Ablate_config.dual_field == True, data.inside == True,  # noqa: E712
            data.bg_sdf == True, vol.volsdf_sampling != False, _DISABLE,  # noqa: E712
            tuple(float(v) for v in data.bound_min), tuple(float(v) for v in data.bound_max), float(vol.rescale),
            float(opt.SDF.NN_Init.scale_mlp), float(data.bg_rad), tuple(float(v) for v in renderer.bg_host),
            float(sdf_field.beta_speed))


def _plan(renderer, opt, sdf_field, rad_field) -> _Plan:
    plans = renderer.__dict__.get("_ls2fm_plans")
    if plans is None:
        plans = renderer.__dict__["_ls2fm_plans"] = {}
    ident = (id(sdf_field), id(rad_field))
    key = _plan_key(renderer, opt, sdf_field, rad_field)
    pl = plans.get(ident)
    if pl is not None and pl.key == key and pl.sdf_ref() is sdf_field and pl.rad_ref() is rad_field:
        if not pl.ok:
            return pl
        ts, _ = param_tensors(sdf_field, rad_field)
        _, g1, g2, dual, _, _ = pl.cfg
        if len(ts) == len(pl.ts) and all(a is b for a, b in zip(ts, pl.ts)) and \
                g1 is sdf_field.embed_fn.embedder_obj.desc and (not dual or g2 is rad_field.embed_fn.embedder_obj.desc):
            return pl
    pl = _Plan()
    pl.key = key

    def _drop(_ref, plans=plans, ident=ident):
        plans.pop(ident, None)
    pl.sdf_ref = weakref.ref(sdf_field, _drop)
    pl.rad_ref = weakref.ref(rad_field, _drop)
    pl.ok = supported(opt, sdf_field, rad_field)
    pl.cfg = pl.ts = pl.pkey = pl.pstruct = pl.dual_table = pl.dual_key = None
    pl.ws_bytes = {}
    if pl.ok:
        ts, dual = param_tensors(sdf_field, rad_field)
        g1 = sdf_field.embed_fn.embedder_obj.desc
        g2 = rad_field.embed_fn.embedder_obj.desc if dual else g1
        pl.cfg = (field_desc(opt, renderer), g1, g2, dual, float(sdf_field.beta_speed), pl)
        pl.ts = ts
    plans[ident] = pl
    return pl


def render_plan(renderer, opt, center, ray, sdf_field, rad_field):
    """the cached plan when the fused kernels serve this call, else None (one plan check per call: pass it on to render())"""
    if not (center.is_cuda and ray.is_cuda):
        return None
    pl = _plan(renderer, opt, sdf_field, rad_field)
    if not pl.ok:
        return None
    n_points = center.shape[0] * center.shape[1] * int(opt.SDF.VolSDF.sample_intvs) if center.dim() == 3 else 0
    if n_points > _lib.MAX_RENDER_POINTS:       # 32-bit offsets inside one call: bigger batches take the composed form
        return None
    return pl                   # pose gradients (center / ray requiring grad) are part of the fused backward


def can_render(renderer, opt, center, ray, sdf_field, rad_field) -> bool:
    return render_plan(renderer, opt, center, ray, sdf_field, rad_field) is not None


_HAS_SDF_EVAL = True


# ------------------------------------------------------------------------------------------------ descriptors
def field_desc(opt, renderer=None) -> _lib.FieldDesc:
    f = _lib.FieldDesc()
    for d in range(3):
        f.bound_min[d] = float(opt.data.bound_min[d])
        f.bound_max[d] = float(opt.data.bound_max[d])
    f.rescale = float(opt.SDF.VolSDF.rescale)
    f.scale_mlp = float(opt.SDF.NN_Init.scale_mlp)
    f.inside = 1 if opt.data.inside == True else 0  # noqa: E712
    f.bg_sdf = 1 if (opt.data.inside == True and opt.data.bg_sdf == True) else 0  # noqa: E712
    f.bg_rad = float(opt.data.bg_rad)
    bg = renderer.bg_host if renderer is not None else [0.0, 0.0, 0.0]        # host copy: no device sync per call
    for d in range(3):
        f.bgcolor[d] = float(bg[d])
    f.n_samples = int(opt.SDF.VolSDF.sample_intvs)
    f.dual_field = 1 if opt.Ablate_config.dual_field == True else 0  # noqa: E712
    return f


def _linear_tensors(layer):
    return [layer.weight_v, layer.weight_g, layer.bias]


def param_tensors(sdf_field, rad_field):
    """flat, fixed-order list of the Parameters the kernels read (order == ls2fm_params)"""
    ts = [sdf_field.embed_fn.embedder_obj.params]
    ts += _linear_tensors(sdf_field.SDF_MLP.mlp[0]) + _linear_tensors(sdf_field.SDF_MLP.mlp[1])
    ts += [sdf_field.beta]
    dual = rad_field is not None and hasattr(rad_field, "Geo_enc")
    if dual:
        ts += [rad_field.embed_fn.embedder_obj.params]
        ts += _linear_tensors(rad_field.Geo_enc.mlp[0]) + _linear_tensors(rad_field.Geo_enc.mlp[1])
    if rad_field is not None:
        for layer in rad_field.Rad_dec.mlp_radiance:
            ts += _linear_tensors(layer)
    return ts, dual


def _fill_linear(dst, ts, at):
    dst.weight_v, dst.weight_g, dst.bias = ptr(ts[at]), ptr(ts[at + 1]), ptr(ts[at + 2])
    return at + 3


def _params_struct(ts, dual, beta_speed, with_rad=True, cls=_lib.Params):
    """ls2fm_params / ls2fm_param_grads over a tensor list laid out like param_tensors()"""
    s = cls()
    at = 0
    s.sdf_table = ptr(ts[at]); at += 1
    at = _fill_linear(s.sdf_mlp[0], ts, at)
    at = _fill_linear(s.sdf_mlp[1], ts, at)
    s.beta = ptr(ts[at]); at += 1
    if cls is _lib.Params:
        s.beta_speed = float(beta_speed)
    if dual:
        s.rad_table = ptr(ts[at]); at += 1
        at = _fill_linear(s.geo_mlp[0], ts, at)
        at = _fill_linear(s.geo_mlp[1], ts, at)
    if with_rad:
        for li in range(3):
            at = _fill_linear(s.rad_mlp[li], ts, at)
    return s


# ------------------------------------------------------------------------------------------------ render
def flat_gradient_views(ps, attach=True):
    """One flat fp32 buffer holding a gradient for each tensor of `ps`, every segment starting on a 16-byte boundary
    (the table scatter stores float4: csrc/bin_scatter.hip) -> (flat, [views shaped like ps]).  A multi-GPU run
    all-reduces `flat` as a single message without packing kernels: each tensor of `ps` gets the buffer attached as
    its `_ls2fm_grad_flat` attribute (it travels with the Parameters whose gradients it holds -- no module state), which is
    where ls2fm.dist.GradAllReducer looks for it."""
    offs, at = [], 0
    for p in ps:
        offs.append(at)
        at += (p.numel() + 3) // 4 * 4
    # a sharded optimizer (ls2fm.dist.ShardedAdam) reduce-scatters this buffer: it then has that optimizer's padded length
    total = max(at, int(getattr(ps[0], "_ls2fm_flat_total", 0)))
    flat = torch.empty(total, device=ps[0].device, dtype=torch.float32)
    if total > at:
        flat[at:].zero_()
    views = [flat[o:o + p.numel()] if p.dim() == 1 else flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, ps)]
    if attach:
        for p in ps:
            p._ls2fm_grad_flat = flat
    return flat, views


# ---- one gradient buffer per backward pass.  A render's backward publishes its gradient views for the duration of the autograd
# pass it runs in; a point-query node of the SAME pass that runs after it (its forward came first: the loops' early key-point
# tracings and point-side terms) adds its contribution into those views in place (ls2fm_sdf_points_bwd_add) and hands autograd
# no gradient of its own -- instead of a second dense table gradient, a second set of MLP gradients and a sum kernel for each.
# OPT-IN per backward pass (`pass_gradient_sharing`): the in-place adds are only seen by autograd if the published buffer is
# still what it holds for the parameter when the adding node runs -- true when every OTHER gradient producer of these parameters
# in the pass runs after the render too or shares as well; a node that ran BEFORE the render has already made autograd sum into
# a tensor of its own, and later in-place adds into the render's buffer would be lost.  ls2fm.stage's loops know their graphs
# (all extra producers are issued ahead of the render) and switch it on around their backward; nothing else does.
_PASS = {}
_PASS_ARMED = [False]
_PASS_SHARING = [False]


class pass_gradient_sharing:
    def __init__(self, on=True):
        self.on = bool(on) and os.environ.get("LS2FM_SHARE_GRADS", "1") != "0"

    def __enter__(self):
        self.prev, _PASS_SHARING[0] = _PASS_SHARING[0], self.on

    def __exit__(self, *exc):
        _PASS_SHARING[0] = self.prev
        _clear_pass()


def _clear_pass():
    _PASS.clear()
    _PASS_ARMED[0] = False


def _publish_pass_gradients(ps, flat):
    """flat: the buffer `flat_gradient_views(ps)` returned.  (Only the buffer is remembered, not the views handed to autograd: a
    second reference to a view keeps AccumulateGrad from taking it as .grad -- it would copy every gradient instead.)"""
    if not _PASS_SHARING[0]:
        return
    _PASS[ps[0].data_ptr()] = (list(ps), flat)
    if not _PASS_ARMED[0]:
        torch.autograd.Variable._execution_engine.queue_callback(_clear_pass)        # runs when this backward pass ends
        _PASS_ARMED[0] = True


def _pass_gradients(ps):
    """the views an earlier node of this backward pass published for these very parameters, or None"""
    entry = _PASS.get(ps[0].data_ptr())
    if entry is None or len(entry[0]) < len(ps) or not all(a.data_ptr() == b.data_ptr() and a.shape == b.shape for a, b in zip(ps, entry[0])):
        return None
    flat, views, at = entry[1], [], 0
    for p in ps:                                       # the segment layout of flat_gradient_views
        views.append(flat[at:at + p.numel()] if p.dim() == 1 else flat[at:at + p.numel()].view(p.shape))
        at += (p.numel() + 3) // 4 * 4
    return views


def _is_table(p) -> bool:
    return p.dim() == 1 and p.numel() > 4096


_RAD_TABLE_AT = 8        # param_tensors(): sdf table, 2 x (v, g, b), beta, rad table, ...


def _refresh_dual_table(lib, plan, sdf_table, rad_table):
    """make the interleaved copy current (see _DUAL_TABLE) and point the parameter struct at it"""
    if getattr(sdf_table, "_ls2fm_no_mirror", False) or getattr(rad_table, "_ls2fm_no_mirror", False):
        plan.pstruct.dual_table = None          # tables rewritten by something that cannot keep a copy current (sharded optimizer)
        return
    if not (sdf_table.dim() == 1 and rad_table.dim() == 1 and sdf_table.numel() == rad_table.numel()):
        raise RuntimeError("ls2fm: dual-field tables of different size")
    mir = getattr(sdf_table, "_ls2fm_mirror", None)
    mir = mir[0] if mir is not None else None
    if mir is None or getattr(rad_table, "_ls2fm_mirror", (None, 1))[0] is not mir:
        mir = TableMirror()
        sdf_table._ls2fm_mirror = (mir, 0)      # (record, which half of every 16-byte entry)
        rad_table._ls2fm_mirror = (mir, 1)
    if mir.table is None or mir.table.numel() != 2 * sdf_table.numel() or mir.table.device != sdf_table.device:
        mir.table = torch.empty(2 * sdf_table.numel(), device=sdf_table.device, dtype=torch.float32)
        mir.versions = [None, None]
    capturing = torch.cuda.is_current_stream_capturing()
    # A captured step cannot re-check versions at replay, so by default the rebuild is recorded into the graph (every replay
    # then renders the CURRENT tables whoever changed them).  The owner of a captured loop whose only table writer is the
    # mirrored optimizer INSIDE the captured step -- or that never changes the tables -- says so with trust_mirror_in_capture():
    # the copy is then current at every replay by construction and the graph holds no rebuild.
    trusted = getattr(sdf_table, "_ls2fm_mirror_trusted", False) and getattr(rad_table, "_ls2fm_mirror_trusted", False)
    keep = mir.fresh(sdf_table, rad_table) and _DUAL_TABLE != "always" and (not capturing or trusted)
    if not keep:
        check(lib.ls2fm_interleave_tables(ptr(sdf_table), ptr(rad_table), sdf_table.numel() // 2, ptr(mir.table), stream_ptr()),
              "ls2fm_interleave_tables")
        mir.ptrs = [sdf_table.data_ptr(), rad_table.data_ptr()]
        mir.versions = [sdf_table._version, rad_table._version]
    plan.dual_table = mir.table
    plan.pstruct.dual_table = ptr(mir.table)


def table_params(sdf_field, rad_field):
    """the two hash-table Parameters of a dual-field pair, or None"""
    if rad_field is None or not hasattr(rad_field, "embed_fn"):
        return None
    return sdf_field.embed_fn.embedder_obj.params, rad_field.embed_fn.embedder_obj.params


def trust_mirror_in_capture(sdf_field, rad_field, on=True):
    """Promise for hipGraph captures of steps over these fields: between replays the hash tables are only ever changed by a
    FusedAdam update that is part of the captured step itself (or not at all), so the interleaved copy needs no rebuild inside
    the graph.  After changing the tables by other means (load_state_dict, a restore), call sync_mirror()."""
    tabs = table_params(sdf_field, rad_field)
    if tabs is not None:
        for t in tabs:
            t._ls2fm_mirror_trusted = bool(on)


def sync_mirror(sdf_field, rad_field):
    """rebuild the interleaved copy of the two tables now (after a table was rewritten behind a captured step's back)"""
    tabs = table_params(sdf_field, rad_field)
    if tabs is None or _DUAL_TABLE == "off":
        return
    rec = getattr(tabs[0], "_ls2fm_mirror", None)
    if rec is None or rec[0].table is None or getattr(tabs[1], "_ls2fm_mirror", (None,))[0] is not rec[0]:
        return
    mir = rec[0]
    check(_lib.load().ls2fm_interleave_tables(ptr(tabs[0]), ptr(tabs[1]), tabs[0].numel() // 2, ptr(mir.table), stream_ptr()),
          "ls2fm_interleave_tables")
    mir.ptrs = [tabs[0].data_ptr(), tabs[1].data_ptr()]
    mir.versions = [tabs[0]._version, tabs[1]._version]


class FusedLoss:
    """What the fused loss head of one render needs (ls2fm_loss_spec): built by ls2fm.losses.RenderLossHead.spec()."""
    __slots__ = ("weights", "rgb_gt", "mask_eik", "mask_dc", "mask_mse", "global_counts", "psnr", "ready", "depth_node", "flags")

    def __init__(self, weights, rgb_gt, mask_eik=None, mask_dc=None, mask_mse=None, global_counts="allreduce"):
        """mask_eik / mask_mse: a uint8 [R] tensor, None (every ray) or "gt": CameraSet.render's mask_bg evaluated in the
        kernels from rgb_gt (LS2FM_LOSS_*_FROM_GT) -- nothing but the traced depth and mask_dc then comes out of a tracing"""
        self.weights, self.rgb_gt = weights, rgb_gt
        self.flags = (_lib.LOSS_EIK_FROM_GT if isinstance(mask_eik, str) else 0) | (_lib.LOSS_MSE_FROM_GT if isinstance(mask_mse, str) else 0)
        mask_eik = None if isinstance(mask_eik, str) else mask_eik
        mask_mse = None if isinstance(mask_mse, str) else mask_mse
        self.mask_eik, self.mask_dc, self.mask_mse = mask_eik, mask_dc, mask_mse
        self.global_counts = global_counts
        self.psnr = None                        # set by the render: terms[6] = -10 log10(mse)
        self.ready = None                       # torch.cuda.Event: masks / traced depth final (produced on another stream)
        self.depth_node = None                  # TracedDepthNode whose d_pred is this render's depth_ref: its backward is run BY
                                                # the render's backward, on another stream, beside the scatter (see _Render.backward)


def _loss_struct(fl, depth_ref, terms, sums, d_terms=None, d_total=None, d_depth_ref=None):
    s = _lib.LossSpec()
    s.rgb_gt, s.depth_ref = ptr(fl.rgb_gt), ptr(depth_ref)
    s.mask_eik, s.mask_dc, s.mask_mse = ptr(fl.mask_eik), ptr(fl.mask_dc), ptr(fl.mask_mse)
    s.weights, s.terms, s.sums = ptr(fl.weights), ptr(terms), ptr(sums)
    s.d_terms, s.d_total, s.d_depth_ref = ptr(d_terms), ptr(d_total), ptr(d_depth_ref)
    s.flags = fl.flags
    s.count_scale = _uniform_count_scale(fl)
    return s


def _uniform_count_scale(fl) -> int:
    """world size when the loss head's global counts are "uniform" under a process group (the forward's reduction then multiplies
    its counts itself: no collective, no kernel between forward and backward), else 0"""
    from . import dist as _dist
    if getattr(fl, "global_counts", None) == "uniform" and _dist.is_distributed():
        return int(_dist.world_size())
    return 0


class _Render(torch.autograd.Function):
    """Renderer.forward (and, with a FusedLoss, the loss head of the stage loops inside it) as ONE autograd node.
    outputs: rgb, sdfs_volume, normals, depth_mlp, normal_mlp [, terms (5), total]"""

    @staticmethod
    def forward(ctx, center, ray, depth_ref, cfg, fl, want_bwd, *params):
        fdesc, g1, g2, dual, beta_speed, plan = cfg
        lib = _lib.load()
        ctx.set_materialize_grads(False)        # unused outputs reach backward as None -> NULL upstream, no zero fills
        dev = center.device
        shape2 = tuple(center.shape[:2])
        n_rays = shape2[0] * shape2[1]
        n = fdesc.n_samples
        c = center.detach().reshape(-1, 3).float().contiguous()
        d = ray.detach().reshape(-1, 3).float().contiguous()
        ps = params                             # nn.Parameters: contiguous fp32 by construction (checked in the plan key)
        pkey = tuple([p.data_ptr() for p in ps])
        if plan.pkey != pkey:                   # pointers only change when a module is re-created / moved
            for p in ps:
                if not (p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda):
                    raise RuntimeError("ls2fm: fused render needs contiguous fp32 GPU parameters")
            plan.pstruct = _params_struct(ps, dual, beta_speed)
            plan.pkey = pkey
            plan.dual_key = None
        if dual and _DUAL_TABLE != "off":
            _refresh_dual_table(lib, plan, ps[0], ps[_RAD_TABLE_AT])
        elif plan.pstruct.dual_table:
            plan.pstruct.dual_table = None      # mode switched off after use: gather from the two tables again
            plan.dual_key = None
        ws_bytes = plan.ws_bytes.get(n_rays)
        if ws_bytes is None:
            ws_bytes = lib.ls2fm_render_workspace_bytes(ctypes.byref(fdesc), ctypes.byref(g1), n_rays)
            if ws_bytes < 0:
                check(int(ws_bytes), "ls2fm_render_workspace_bytes")
            plan.ws_bytes[n_rays] = ws_bytes
        ws = torch.empty(ws_bytes // 4, device=dev, dtype=torch.float32)
        if _POISON:
            ws.fill_(float("nan"))          # debug: any read of a workspace word that was never written shows up as NaN
        rgb = torch.empty(*shape2, 3, device=dev)
        sdfs = torch.empty(*shape2, n, 1, device=dev)
        normals = torch.empty(*shape2, n, 3, device=dev)
        depth = torch.empty(*shape2, 1, device=dev)
        nmlp = torch.empty(*shape2, 3, device=dev)
        pstruct = plan.pstruct
        opts = _lib.RenderOpts()
        opts.inference_only = 0 if want_bwd else 1
        terms = sums = dref = None
        if fl is not None:
            dref = None if depth_ref is None else depth_ref.detach().reshape(-1).float().contiguous()
            if fl.rgb_gt.numel() != 3 * n_rays or (dref is not None and dref.numel() != n_rays):
                raise RuntimeError("ls2fm: fused loss head: rgb_gt / d_points do not match the rays")
            terms = torch.empty(8, device=dev, dtype=torch.float32)      # rgb, eikonal, DC, mse, all, all, PSNR, -
            sums = torch.empty(8, device=dev, dtype=torch.float64)
            lspec = _loss_struct(fl, dref, terms, sums)
            opts.loss = ctypes.pointer(lspec)
            if fl.ready is not None:
                opts.loss_inputs_ready = fl.ready.cuda_event
        check(lib.ls2fm_render_fwd(ctypes.byref(fdesc), ctypes.byref(g1), ctypes.byref(g2) if dual else None,
                                   ctypes.byref(pstruct), ptr(c), ptr(d), n_rays, ptr(rgb), ptr(sdfs), ptr(normals),
                                   ptr(depth), ptr(nmlp), ptr(ws), ctypes.byref(opts), stream_ptr()), "ls2fm_render_fwd")
        if fl is not None:
            from . import dist as _dist
            if _dist.is_distributed() and lspec.count_scale <= 1:       # world-size-invariant means: global counts (and sums) before the backward
                _dist.globalize_loss_sums(sums, fl.global_counts)
                check(lib.ls2fm_loss_terms_from_sums(ptr(sums), ptr(fl.weights), ptr(terms), stream_ptr()),
                      "ls2fm_loss_terms_from_sums")
        ctx.cfg = cfg
        ctx.ws = ws
        ctx.n_rays = n_rays
        ctx.pstruct = pstruct
        ctx.pose_shape = tuple(center.shape)
        ctx.loss = (fl, dref, sums, None if depth_ref is None else tuple(depth_ref.shape))
        ctx.save_for_backward(c, d, *ps)
        if fl is None:
            return rgb, sdfs, normals, depth, nmlp
        fl.psnr = terms[6]                      # (no graph: the stages only log it)
        return rgb, sdfs, normals, depth, nmlp, terms[:5], terms[5]

    @staticmethod
    def backward(ctx, d_rgb, d_sdfs, d_normals, d_depth, d_nmlp, d_terms=None, d_total=None):
        fdesc, g1, g2, dual, beta_speed, _ = ctx.cfg
        lib = _lib.load()
        c, d, *ps = ctx.saved_tensors

        def prep(t):
            return None if t is None else t.float().contiguous()
        d_rgb, d_sdfs, d_normals, d_depth, d_nmlp, d_terms, d_total = map(
            prep, (d_rgb, d_sdfs, d_normals, d_depth, d_nmlp, d_terms, d_total))
        # every gradient -- both tables (overwritten in full by the slab scatter) and the small tensors -- is a view of ONE
        # flat buffer: a multi-GPU run all-reduces it as a single message without packing kernels (ls2fm.dist)
        flat, grads = flat_gradient_views(ps)
        pstruct = ctx.pstruct
        gstruct = _params_struct(grads, dual, beta_speed, cls=_lib.ParamGrads)
        want_pose = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        d_center = torch.empty_like(c) if want_pose else None
        d_ray = torch.empty_like(d) if want_pose else None
        fl, dref, sums, dref_shape = ctx.loss
        opts = _lib.RenderOpts()
        d_dref = None
        # multi-GPU: scatter the levels in groups and all-reduce a group's table slices while the next group is scattered
        from . import dist as _dist
        n_groups = int(getattr(ps[0], "_ls2fm_overlap_groups", 0)) if _dist.is_distributed() else 0
        if n_groups > 1 and torch.cuda.is_current_stream_capturing() and not (
                _dist.capture_overlap_enabled() and getattr(ps[0], "_ls2fm_group_exchange", None) is None):
            # captured steps: a pipelined sharded optimizer issues its chain at step(), a GradAllReducer reduces the flat buffer
            # as one message on the capturing stream (ls2fm.dist).  Rounds 2 - 5 refused the level-group overlap inside a capture
            # outright: the render's backward then forked a side stream of its own, and a communication-stream branch that forks
            # again into RCCL's stream was the depth-three fork tree that takes hipStreamEndCapture down on ROCm 7.2.  Since
            # round 5 the backward is ONE chain on the capturing stream (csrc/side_jobs.h), so capturing stream -> communication
            # stream -> RCCL's stream is a tree of depth two: ls2fm.dist.enable_capture_overlap() / LS2FM_CAPTURE_OVERLAP=1 records
            # the first group's all-reduce on a second branch of the graph, beside the scatter of the later groups.
            n_groups = 0
        if n_groups > 1 and getattr(ps[0], "_ls2fm_group_exchange", None) is not None and fl is not None and fl.depth_node is not None:
            # a traced-depth node rides in this backward (ls2fm_depth_backward adds into the tables BEHIND the scatter: a group's
            # slices would not be final at its event): a pipelined sharded optimizer then exchanges at its step() instead
            n_groups = 0
        events = []
        if n_groups > 1:
            n_groups = min(n_groups, 4, g1.n_levels)
            events = getattr(ps[0], "_ls2fm_group_events", None)      # kept with the table Parameter: re-recorded every step
            if events is None or len(events) != n_groups:
                cur = torch.cuda.current_stream()
                events = [torch.cuda.Event() for _ in range(n_groups)]
                for ev in events:
                    ev.record(cur)                 # materialises the handle (torch creates events lazily)
                ps[0]._ls2fm_group_events = events
            for gi, ev in enumerate(events):
                opts.group_events[gi] = ev.cuda_event
            opts.n_level_groups = n_groups
        node = fl.depth_node if fl is not None else None
        keep = None
        if fl is not None and (d_terms is not None or d_total is not None):
            if dref is not None and (ctx.needs_input_grad[2] or node is not None):
                d_dref = torch.empty_like(dref)
            if node is not None and d_dref is not None:
                # the traced depth's own backward (20 k track points through the point-query machinery) runs INSIDE this call,
                # merged into its chains (ls2fm_depth_backward): the points' rows join the SDF weight-gradient kernel as extra
                # tiles, their table gradient is added into this node's table behind its scatter -- autograd sees ONE producer
                # per parameter and launches no accumulation kernels of its own
                keep = node.prepare()
                opts.depth_bwd = ctypes.pointer(keep[0])
            lspec = _loss_struct(fl, dref, None, sums, d_terms, d_total, d_dref)
            opts.loss = ctypes.pointer(lspec)
        check(lib.ls2fm_render_bwd(ctypes.byref(fdesc), ctypes.byref(g1), ctypes.byref(g2) if dual else None,
                                   ctypes.byref(pstruct), ptr(c), ptr(d), ctx.n_rays, ptr(d_rgb), ptr(d_sdfs),
                                   ptr(d_normals), ptr(d_depth), ptr(d_nmlp), ctypes.byref(gstruct), ptr(d_center), ptr(d_ray),
                                   ptr(ctx.ws), ctypes.byref(opts), stream_ptr()), "ls2fm_render_bwd")
        if keep is not None:
            d_dref = None                              # consumed inside the call: the tracing node gets no gradient through autograd
        exchanging = False
        if events:
            tables = [grads[0]] + ([grads[_RAD_TABLE_AT]] if dual else [])
            # a pipelined sharded optimizer takes the groups itself (reduce-scatter -> Adam -> all-gather per group); else all-reduces.
            # A hook that DECLINES (another layout, a capture) exchanges everything at its step(): nothing is launched here then
            # (round-4 advisor: falling through to the all-reduce form summed the gradients twice)
            hook = getattr(ps[0], "_ls2fm_group_exchange", None)
            if hook is None:
                _dist.launch_group_reductions(flat, tables, list(g1.offset), events, g1.n_levels, owners=ps)
                exchanging = True
            else:
                exchanging = bool(hook(flat, tables, list(g1.offset), events, g1.n_levels))
        if not exchanging:
            _publish_pass_gradients(ps, flat)          # (level groups: reductions of `flat` are already in flight -- no late adds)
        if want_pose:
            d_center, d_ray = d_center.view(ctx.pose_shape), d_ray.view(ctx.pose_shape)
        if d_dref is not None:
            d_dref = d_dref.view(dref_shape)
        return (d_center, d_ray, d_dref, None, None, None, *grads)  # ctx.ws is kept: backward may run again (retain_graph)


def render(renderer, opt, center, ray, sdf_field, rad_field, loss=None, d_points=None, plan=None):
    """Renderer.forward through the fused kernels -> the reference's result dict.  With `loss` (a FusedLoss) the loss head
    runs inside the render and the dict also holds 'loss_terms' ([5]: rgb, eikonal, DC, mse, all) and 'loss_total'.
    `plan`: what render_plan() just returned for the same arguments (saves the second check)."""
    pl = plan if plan is not None else _plan(renderer, opt, sdf_field, rad_field)
    want_bwd = torch.is_grad_enabled() and (center.requires_grad or ray.requires_grad or any(p.requires_grad for p in pl.ts)
                                            or (d_points is not None and d_points.requires_grad))
    if loss is not None and loss.depth_node is not None and d_points is not None:
        # the render's backward runs the tracing node's backward itself (beside its scatter): no autograd edge to d_points
        # (not with the level-group overlap of a multi-GPU run: a group's table slices must be final at its event)
        from . import dist as _dist
        grouped = _dist.is_distributed() and int(getattr(pl.ts[0], "_ls2fm_overlap_groups", 0)) > 1
        if grouped or not (loss.depth_node.matches(pl.ts) and d_points.requires_grad):
            loss.depth_node = None
        else:
            d_points = d_points.detach()
    out = _Render.apply(center, ray, d_points if loss is not None else None, pl.cfg, loss, want_bwd, *pl.ts)
    ret = {"rgb": out[0], "sdfs_volume": out[1], "normals": out[2], "depth_mlp": out[3], "normal_mlp": out[4]}
    if loss is not None:
        ret["loss_terms"], ret["loss_total"] = out[5], out[6]
        ret["loss_psnr"] = loss.psnr.detach()
    return ret


# ------------------------------------------------------------------------------------------------ sdf eval / tracing
def _sdf_only_params(sdf_field):
    ts, _ = param_tensors(sdf_field, None)
    ts = [t.detach().contiguous() for t in ts]
    return ts, _params_struct(ts, False, float(sdf_field.beta_speed), with_rad=False)


def _sdf_workspace(dev):
    n = _lib.load().ls2fm_sdf_eval_workspace_bytes()
    return torch.empty(n // 4, device=dev, dtype=torch.float32)


def sdf_eval(sdf_field, xyz, want_feat=False, want_normal=False):
    """no-graph SDF.infer_sdf: xyz [...,3] -> sdf [...,1], feat [...,17] | None (, normal [...,3])"""
    lib = _lib.load()
    shape = tuple(xyz.shape[:-1])
    p = xyz.detach().reshape(-1, 3).float().contiguous()
    n = p.shape[0]
    dev = p.device
    sdf = torch.empty(n, device=dev)
    feat = torch.empty(n, _lib.FEAT + 1, device=dev) if want_feat else None
    normal = torch.empty(n, 3, device=dev) if want_normal else None
    keep, pstruct = _sdf_only_params(sdf_field)
    fdesc = field_desc(sdf_field.opt)
    ws = _sdf_workspace(dev)
    check(lib.ls2fm_sdf_eval(ctypes.byref(fdesc), ctypes.byref(sdf_field.embed_fn.embedder_obj.desc), ctypes.byref(pstruct),
                             ptr(p), n, ptr(sdf), ptr(feat), ptr(normal), ptr(ws), stream_ptr()), "ls2fm_sdf_eval")
    out = (sdf.view(*shape, 1), feat.view(*shape, -1) if want_feat else None)
    return out + (normal.view(*shape, 3),) if want_normal else out


# ------------------------------------------------------------------------------------------------ point queries with a graph
def can_query_points(sdf_field, xyz) -> bool:
    """grad-enabled SDF.infer_sdf / SDF.gradient / get_surface_pts through ONE fused node (ls2fm_sdf_eval forward,
    ls2fm_sdf_points_bwd backward); the background-sphere min and non-reference layer sizes take the composed form"""
    if not (xyz.is_cuda and supported(sdf_field.opt, sdf_field)):
        return False
    opt = sdf_field.opt
    if opt.data.inside == True and opt.data.bg_sdf == True:  # noqa: E712
        return False
    n = xyz.numel() // 3
    return 0 < n <= _lib.MAX_RENDER_POINTS


class _SdfPoints(torch.autograd.Function):
    """xyz [...,3] -> sdf [...,1], feat [...,17], normal [...,3] (the analytic d sdf / d xyz).  All three are differentiable
    outputs: the backward handles the normal's upstream analytically (the double backward of SDF.gradient's
    autograd.grad(create_graph=True)), so callers may put norms of `normal` inside losses."""

    @staticmethod
    def forward(ctx, xyz, sdf_field, want_feat, want_normal, *params):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        shape = tuple(xyz.shape[:-1])
        p = xyz.detach().reshape(-1, 3).float().contiguous()
        n = p.shape[0]
        dev = p.device
        fdesc = field_desc(sdf_field.opt)
        gdesc = sdf_field.embed_fn.embedder_obj.desc
        for t in params:
            if not (t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda):
                raise RuntimeError("ls2fm: fused point query needs contiguous fp32 GPU parameters")
        pstruct = _params_struct(list(params), False, float(sdf_field.beta_speed), with_rad=False)
        sdf = torch.empty(n, device=dev)
        feat = torch.empty(n, _lib.FEAT + 1, device=dev) if want_feat else None
        normal = torch.empty(n, 3, device=dev) if want_normal else None
        ws = _sdf_workspace(dev)
        check(lib.ls2fm_sdf_eval(ctypes.byref(fdesc), ctypes.byref(gdesc), ctypes.byref(pstruct), ptr(p), n, ptr(sdf), ptr(feat),
                                 ptr(normal), ptr(ws), stream_ptr()), "ls2fm_sdf_eval")
        ctx.meta = (fdesc, gdesc, float(sdf_field.beta_speed), n, tuple(xyz.shape))
        ctx.save_for_backward(p, *params)
        z = p.new_empty(())                        # placeholder of an output nobody asked for (never read): no fill kernel
        return (sdf.view(*shape, 1), feat.view(*shape, -1) if want_feat else z, normal.view(*shape, 3) if want_normal else z)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_sdf, d_feat, d_normal):
        lib = _lib.load()
        fdesc, gdesc, beta_speed, n, xyz_shape = ctx.meta
        p, *ps = ctx.saved_tensors
        if d_sdf is None and d_feat is None and d_normal is None:
            return (None,) * (4 + len(ps))

        def prep(t, width):
            return None if t is None else t.reshape(-1, width).float().contiguous()
        d_sdf, d_feat, d_normal = prep(d_sdf, 1), prep(d_feat, _lib.FEAT + 1), prep(d_normal, 3)
        shared = _pass_gradients(ps)                   # an earlier node of this pass owns the gradient buffer: add into it
        flat, grads = (None, shared) if shared is not None else flat_gradient_views(ps)          # table, (v, g, b) x 2, beta
        gstruct = _params_struct(grads, False, beta_speed, with_rad=False, cls=_lib.ParamGrads)
        pstruct = _params_struct(ps, False, beta_speed, with_rad=False)
        d_p = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        ws_bytes = lib.ls2fm_sdf_points_workspace_bytes(ctypes.byref(fdesc), ctypes.byref(gdesc), n)
        if ws_bytes < 0:
            check(int(ws_bytes), "ls2fm_sdf_points_workspace_bytes")
        ws = torch.empty(ws_bytes // 4, device=p.device, dtype=torch.float32)
        fn = lib.ls2fm_sdf_points_bwd_add if shared is not None else lib.ls2fm_sdf_points_bwd
        check(fn(ctypes.byref(fdesc), ctypes.byref(gdesc), ctypes.byref(pstruct), ptr(p), n, ptr(d_sdf),
                 ptr(d_feat), ptr(d_normal), ctypes.byref(gstruct), ptr(d_p), ptr(ws), stream_ptr()), "ls2fm_sdf_points_bwd")
        if shared is not None:
            grads = [None] * len(ps)                   # already inside the pass's gradient buffer
        else:
            grads[7] = None                            # beta does not enter a point query: no gradient (not a zero tensor)
        return (None if d_p is None else d_p.view(xyz_shape), None, None, None, *grads)


class _SurfacePts(torch.autograd.Function):
    """SDF.get_surface_pts' projection line as one node each way (ls2fm_surface_pts_fwd / _bwd):
    out = p - n / |n|.detach() * sdf,  length = |n|"""

    @staticmethod
    def forward(ctx, pts, normals, sdf):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        p = pts.detach().reshape(-1, 3).float().contiguous()
        nr = normals.detach().reshape(-1, 3).float().contiguous()
        sd = sdf.detach().reshape(-1).float().contiguous()
        n = p.shape[0]
        out, length = torch.empty_like(p), torch.empty(n, device=p.device)
        check(lib.ls2fm_surface_pts_fwd(ptr(p), ptr(nr), ptr(sd), n, ptr(out), ptr(length), stream_ptr()), "ls2fm_surface_pts_fwd")
        ctx.save_for_backward(nr, sd, length)
        ctx.shapes = (tuple(pts.shape), tuple(normals.shape), tuple(sdf.shape))
        return out.view(pts.shape), length.view(*pts.shape[:-1], 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out, g_len):
        lib = _lib.load()
        nr, sd, length = ctx.saved_tensors
        if g_out is None and g_len is None:
            return None, None, None
        go = None if g_out is None else g_out.reshape(-1, 3).float().contiguous()
        gl = None if g_len is None else g_len.reshape(-1).float().contiguous()
        d_n, d_s = torch.empty_like(nr), torch.empty_like(sd)
        check(lib.ls2fm_surface_pts_bwd(ptr(nr), ptr(sd), ptr(length), nr.shape[0], ptr(go), ptr(gl), ptr(d_n), ptr(d_s), stream_ptr()),
              "ls2fm_surface_pts_bwd")
        return (None if go is None else go.view(ctx.shapes[0])), d_n.view(ctx.shapes[1]), d_s.view(ctx.shapes[2])


def surface_points(pts, normals, sdf):
    """(pts - normals / |normals|.detach() * sdf, |normals|) with a graph, one launch each way"""
    return _SurfacePts.apply(pts, normals, sdf)


def query_points(sdf_field, xyz, want_feat=False, want_normal=False):
    """-> (sdf [...,1], feat [...,17] | None, normal [...,3] | None), graph attached"""
    ts, _ = param_tensors(sdf_field, None)
    sdf, feat, normal = _SdfPoints.apply(xyz, sdf_field, want_feat, want_normal, *ts)
    return sdf, (feat if want_feat else None), (normal if want_normal else None)


class _TracedDepth(torch.autograd.Function):
    """SDF.sphere_tracing's differentiable tail for a whole ray batch as ONE autograd node (ls2fm_sdf_eval on the track points,
    ls2fm_trace_depth_fwd; backward: ls2fm_trace_depth_bwd, ls2fm_sdf_points_bwd), with the trip count K staying on the device.
    -> d_pred [R], sdf_last [R] (differentiable), finish, mask_bg, mask_dc (uint8 [R])"""

    @staticmethod
    def forward(ctx, track, trips, near, far, rgb_gt, finish_thr, sdf_field, trace_ws, launch_stream, *params):
        if launch_stream is not None:          # kernels on that stream (the caller orders it against the current one)
            with torch.cuda.stream(launch_stream):
                outs = _TracedDepth._forward(ctx, track, trips, near, far, rgb_gt, finish_thr, sdf_field, trace_ws, *params)
            if not torch.cuda.is_current_stream_capturing():
                cur = torch.cuda.current_stream(track.device)
                for t in list(outs) + list(ctx.side_allocated):
                    t.record_stream(cur)       # allocated on the launch stream, used (and freed) on the current one
            ctx.side_allocated = None
            return outs
        return _TracedDepth._forward(ctx, track, trips, near, far, rgb_gt, finish_thr, sdf_field, trace_ws, *params)

    @staticmethod
    def _forward(ctx, track, trips, near, far, rgb_gt, finish_thr, sdf_field, trace_ws, *params):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        n_rays, k_max = track.shape[0], track.shape[1]
        p = track.detach().reshape(-1, 3).float().contiguous()
        dev = p.device
        fdesc = field_desc(sdf_field.opt)
        gdesc = sdf_field.embed_fn.embedder_obj.desc
        for t in params:
            if not (t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda):
                raise RuntimeError("ls2fm: fused point query needs contiguous fp32 GPU parameters")
        pstruct = _params_struct(list(params), False, float(sdf_field.beta_speed), with_rad=False)
        sdf = getattr(track, "_ls2fm_track_sdf", None)
        if sdf is not None and sdf.numel() == n_rays * k_max:
            sdf = sdf.reshape(-1)               # the tracing loop's own values at the track points
        elif trace_ws is not None:              # the tracing call's workspace: its packed weights are these parameters'
            sdf = torch.empty(n_rays * k_max, device=dev)
            check(lib.ls2fm_sdf_eval_prepared(ctypes.byref(fdesc), ctypes.byref(gdesc), ctypes.byref(pstruct), ptr(p), n_rays * k_max,
                                              ptr(sdf), ptr(trace_ws), stream_ptr()), "ls2fm_sdf_eval_prepared")
        else:
            sdf = torch.empty(n_rays * k_max, device=dev)
            ws = _sdf_workspace(dev)
            check(lib.ls2fm_sdf_eval(ctypes.byref(fdesc), ctypes.byref(gdesc), ctypes.byref(pstruct), ptr(p), n_rays * k_max,
                                     ptr(sdf), None, None, ptr(ws), stream_ptr()), "ls2fm_sdf_eval")
        d_pred = torch.empty(n_rays, device=dev)
        last = torch.empty(n_rays, device=dev)
        finish, gate = (torch.empty(n_rays, device=dev, dtype=torch.uint8) for _ in range(2))
        gt = None
        mask_bg = mask_dc = None
        if rgb_gt is not None:
            gt = rgb_gt.detach().reshape(-1, 3).float().contiguous()
            if gt.shape[0] != n_rays:
                raise RuntimeError("ls2fm: traced depth: rgbs_gt does not match the rays")
            mask_bg, mask_dc = (torch.empty(n_rays, device=dev, dtype=torch.uint8) for _ in range(2))
        check(lib.ls2fm_trace_depth_fwd(ptr(sdf), ptr(trips), ptr(near), ptr(far), n_rays, k_max, float(finish_thr), ptr(gt), 0.05,
                                        0.95, ptr(d_pred), ptr(last), ptr(finish), ptr(mask_bg), ptr(mask_dc), ptr(gate),
                                        stream_ptr()), "ls2fm_trace_depth_fwd")
        ctx.meta = (fdesc, gdesc, float(sdf_field.beta_speed), n_rays, k_max)
        ctx.side_allocated = [p, gate, sdf]
        ctx.save_for_backward(p, trips, gate, *params)
        sdf_field.last_trace_node = TracedDepthNode(ctx.meta, p, trips, gate, params,
                                                    keep=(track, getattr(track, "_ls2fm_keep", None), near, far, sdf, gt))
        outs = (d_pred, last, finish, mask_bg if mask_bg is not None else finish, mask_dc if mask_dc is not None else finish)
        ctx.mark_non_differentiable(*outs[2:])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_dpred, d_last, *_):
        p, trips, gate, *ps = ctx.saved_tensors
        n_in = 9
        if d_dpred is None and d_last is None:
            return (None,) * (n_in + len(ps))
        _, grads = _traced_depth_backward(ctx.meta, p, trips, gate, ps, d_dpred, d_last, attach=True)
        grads[7] = None                                # beta does not enter a point query (None already when the pass shares a buffer)
        return (None,) * n_in + tuple(grads)


def _traced_depth_backward(meta, p, trips, gate, ps, d_dpred, d_last, attach):
    """upstream of the traced depth / last SDF value -> the SDF field's parameter gradients, through the point-query backward
    -> (flat buffer, views).  attach: register the buffer with the Parameters (the node is an autograd producer of its own)"""
    lib = _lib.load()
    fdesc, gdesc, beta_speed, n_rays, k_max = meta
    d_dpred = None if d_dpred is None else d_dpred.reshape(-1).float().contiguous()
    d_last = None if d_last is None else d_last.reshape(-1).float().contiguous()
    d_sdf = torch.empty(n_rays * k_max, device=p.device)
    check(lib.ls2fm_trace_depth_bwd(ptr(d_dpred), ptr(d_last), ptr(trips), ptr(gate), n_rays, k_max, ptr(d_sdf), stream_ptr()),
          "ls2fm_trace_depth_bwd")
    shared = _pass_gradients(ps) if attach else None
    flat, grads = (None, shared) if shared is not None else flat_gradient_views(ps, attach=attach)          # table, (v, g, b) x 2, beta
    gstruct = _params_struct(grads, False, beta_speed, with_rad=False, cls=_lib.ParamGrads)
    pstruct = _params_struct(ps, False, beta_speed, with_rad=False)
    n = n_rays * k_max
    ws_bytes = lib.ls2fm_sdf_points_workspace_bytes(ctypes.byref(fdesc), ctypes.byref(gdesc), n)
    if ws_bytes < 0:
        check(int(ws_bytes), "ls2fm_sdf_points_workspace_bytes")
    ws = torch.empty(ws_bytes // 4, device=p.device, dtype=torch.float32)
    fn = lib.ls2fm_sdf_points_bwd_add if shared is not None else lib.ls2fm_sdf_points_bwd
    check(fn(ctypes.byref(fdesc), ctypes.byref(gdesc), ctypes.byref(pstruct), ptr(p), n, ptr(d_sdf),
             None, None, ctypes.byref(gstruct), None, ptr(ws), stream_ptr()), "ls2fm_sdf_points_bwd")
    return flat, ([None] * len(ps) if shared is not None else list(grads))


class TracedDepthNode:
    """What a render needs to run a traced depth's backward itself (FusedLoss.depth_node): the tensors `_TracedDepth` saved."""

    def __init__(self, meta, p, trips, gate, params, keep=None):
        self.meta, self.p, self.trips, self.gate, self.params = meta, p, trips, gate, list(params)
        # every tensor the tracing's kernels touched on their launch stream: alive as long as this node is the field's last one,
        # i.e. past the caller's join with that stream.  (Under hipGraph capture there is no record_stream: a block freed in
        # Python while the other stream's kernels are still ahead of the join would be handed to the next allocation of the
        # capturing stream -- measured: the render's outputs landing in the tracing's near / far / track values.)
        self.keep = keep

    def matches(self, render_params) -> bool:
        """the tracing's parameters are the first tensors of the render's list (same flat-buffer offsets)"""
        return len(render_params) >= len(self.params) and all(a is b for a, b in zip(self.params, render_params))

    def prepare(self):
        """ls2fm_depth_backward for a render backward -> (struct, tensors to keep alive until the call has been enqueued)"""
        lib = _lib.load()
        fdesc, gdesc, beta_speed, n_rays, k_max = self.meta
        dev = self.p.device
        n = n_rays * k_max
        ws_bytes = lib.ls2fm_sdf_points_workspace_bytes(ctypes.byref(fdesc), ctypes.byref(gdesc), n)
        if ws_bytes < 0:
            check(int(ws_bytes), "ls2fm_sdf_points_workspace_bytes")
        ws = torch.empty(ws_bytes // 4, device=dev, dtype=torch.float32)
        d_sdf = torch.empty(n, device=dev)
        db = _lib.DepthBackward()
        db.points, db.trips, db.gate, db.k_max = ptr(self.p), ptr(self.trips), ptr(self.gate), k_max
        db.d_sdf, db.workspace = ptr(d_sdf), ptr(ws)
        return db, (ws, d_sdf)


def traced_depth(sdf_field, track, trips, near, far, rgbs_gt=None, trace_ws=None, launch_stream=None):
    """track [R, iters_max (+1), 3], trips int32[1] on the device, near / far [R] (what sphere_trace(sync=False) returns) ->
    d_pred [R], sdf_last [R] (graph attached), finish_mask [R] bool, and with rgbs_gt [R,3]: mask_bg, mask_finish & mask_bg as
    uint8 [R] (what the fused loss head takes) -- else None, None"""
    pts = track                                  # all iters_max + 1 columns (a slice would be copied): K <= iters_max masks the rest
    thr = _finish_threshold(sdf_field)
    ts, _ = param_tensors(sdf_field, None)
    d_pred, last, finish, mask_bg, mask_dc = _TracedDepth.apply(pts, trips, near.reshape(-1), far.reshape(-1), rgbs_gt, thr, sdf_field,
                                                                trace_ws, launch_stream, *ts)
    if rgbs_gt is None:
        return d_pred, last, finish.view(torch.bool), None, None
    return d_pred, last, finish.view(torch.bool), mask_bg, mask_dc


def _finish_threshold(sdf_field):
    """(bound_max[0] - bound_min[0]) / 10 / Res in the reference's fp32 arithmetic (SDF.py:213-214), from the options: no device
    round trip (the module's bound tensors live on the GPU)"""
    import numpy as np
    data = sdf_field.opt.data
    extent = np.float32(data.bound_max[0]) - np.float32(data.bound_min[0])
    return float(extent / np.float32(10) / np.float32(sdf_field.opt.Res))


def sdf_volume(sdf_field, n_side, step, origin, first=0, count=None, reference_indexing=True):
    """no-graph SDF sweep over an n_side^3 lattice built on the device (ls2fm_sdf_volume): flat float32 [count].
    step / origin: 3 doubles each, per output column; reference_indexing: the lattice arithmetic of the reference's
    extract_mesh (utils/util.py:399-409), else integer lattice indices."""
    lib = _lib.load()
    dev = sdf_field.beta.device
    if dev.type != "cuda":
        raise RuntimeError("ls2fm: the SDF sweep runs on the GPU only (no CPU fallback)")
    n_side = int(n_side)
    count = n_side ** 3 - first if count is None else int(count)
    out = torch.empty(count, device=dev)
    keep, pstruct = _sdf_only_params(sdf_field)
    fdesc = field_desc(sdf_field.opt)
    ws = _sdf_workspace(dev)
    st = (ctypes.c_double * 3)(*[float(v) for v in step])
    og = (ctypes.c_double * 3)(*[float(v) for v in origin])
    check(lib.ls2fm_sdf_volume(ctypes.byref(fdesc), ctypes.byref(sdf_field.embed_fn.embedder_obj.desc), ctypes.byref(pstruct),
                               n_side, int(first), count, 1 if reference_indexing else 0, st, og, ptr(out), ptr(ws),
                               stream_ptr()), "ls2fm_sdf_volume")
    return out


def sphere_trace(sdf_field, o, d, history=False, sync=True, launch_stream=None):
    """the reference's root-find loop (SDF.py:149-200) in one kernel.
    o, d [R,3] -> near [R], far [R], pts_tracks [R,K,3], t_end [R] (far-end distance after K trips), K
    history=True: t_end comes back as [R,K+1], the far-end distance after every trip (parity tests)
    sync=False: no host round trip -- the full track [R, iters_max + 1, 3], t_end [R, iters_max + 1] and K as a DEVICE int32[1]
    (sharded rays: already max-reduced over the ranks); the caller masks the trips beyond K (SDF.sphere_tracing(static_trips=True))"""
    lib = _lib.load()
    o = o.detach().float().contiguous()
    d = d.detach().float().contiguous()
    n = o.shape[0]
    dev = o.device
    it = int(sdf_field.iters_max)
    near = torch.empty(n, device=dev)
    far = torch.empty(n, device=dev)
    track = torch.empty(n, it + 1, 3, device=dev)
    t_end = torch.empty(n, it + 1, device=dev)
    trips = torch.empty(1, device=dev, dtype=torch.int32)            # zeroed by the call
    # the field's value at every track point, straight from the loop (what evaluating the track afterwards returns, bit for bit):
    # the static form's depth node then needs no evaluation pass of its own
    track_sdf = torch.empty(n, it + 1, device=dev) if not sync else None
    keep, pstruct = _sdf_only_params(sdf_field)
    fdesc = field_desc(sdf_field.opt)
    gdesc = sdf_field.embed_fn.embedder_obj.desc
    ws = _sdf_workspace(dev)
    args = (ctypes.byref(fdesc), ctypes.byref(gdesc), ctypes.byref(pstruct), ptr(o), ptr(d), n, float(sdf_field.sdf_threshold), it,
            ptr(near), ptr(far), ptr(track), ptr(t_end), ptr(track_sdf), ptr(trips), ptr(ws))
    if launch_stream is None:
        check(lib.ls2fm_sphere_trace(*args, stream_ptr()), "ls2fm_sphere_trace")
    else:
        # weight preparation (one latency-bound workgroup) on the CURRENT stream, the root-find on `launch_stream` behind it: the
        # caller runs that stream beside other work of the current one (ls2fm.stage) and orders its readers itself
        check(lib.ls2fm_sdf_prepare(ctypes.byref(gdesc), ctypes.byref(pstruct), ptr(ws), ptr(trips), stream_ptr()), "ls2fm_sdf_prepare")
        launch_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(launch_stream):
            check(lib.ls2fm_sphere_trace_prepared(*args, stream_ptr()), "ls2fm_sphere_trace_prepared")
        if not torch.cuda.is_current_stream_capturing():
            for t in (near, far, track, t_end, trips, ws, track_sdf):
                if t is not None:
                    t.record_stream(launch_stream)
    from . import dist as _dist
    if not sync:
        if _dist.is_distributed():
            import torch.distributed as tdist
            import contextlib
            with (torch.cuda.stream(launch_stream) if launch_stream is not None else contextlib.nullcontext()):
                tdist.all_reduce(trips, op=tdist.ReduceOp.MAX)
        track._ls2fm_trace_ws = ws              # packed weights of these parameters: reused by traced_depth (no second prep)
        track._ls2fm_track_sdf = track_sdf
        track._ls2fm_keep = (o, d, near, far, t_end, trips, track_sdf, ws)
        return near, far, track, t_end, trips
    k = int(trips.item())           # the reference syncs here too (its loop condition is a host-side .sum())
    k = _dist.global_max_int(k, dev)                  # sharded rays: keep K identical to the single-GPU run
    pts = track[:, :max(k, 1), :]                     # K == 0: the single current point (SDF.py:201-202)
    return near, far, pts, (t_end[:, :k + 1] if history else t_end[:, k]), k
