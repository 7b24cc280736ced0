"""Loss head over the renderer's outputs -- the host-side mirror of what the reference's stages compute right after
``Renderer.forward``:

    rgb_loss      = l1_loss(rgb, rgbs_gt)                                          pipelines/Camera.py:535
    eikonal_loss  = l1_loss(norm(normals[mask], dim=-1), 1)                        Initialization.py:257-258, BA.py:193-194
    DC_loss       = smooth_l1_loss(d_points[mask_finish], depth_mlp[mask_finish])  Camera.py:520-532 (0 for an empty mask)
    PSNR          = -10 log10 mse_loss(rgb[mask_bg], rgbs_gt[mask_bg])             Camera.py:533
    all           = sum_k 10**w_k * loss_k                                         BA.py:206-218 (summarize_loss)

One HIP kernel forward, one backward (csrc/loss_head.hip) instead of ~35 launch-bound elementwise / reduction kernels;
the sums are deterministic (fixed-order fp64 partials).  No CPU fallback.
"""
from __future__ import annotations

import torch

from . import _lib

_WS = {}
_LAST_SUMS = {}


def _workspace(dev):
    """one zero-filled workspace per (device, stream): the kernel re-arms its ticket itself"""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.zeros(_lib.load().ls2fm_loss_head_workspace_bytes() // 8, device=dev, dtype=torch.float64)
        _WS[key] = ws
    return ws


def bg_mask(rgbs_gt):
    """CameraSet.render's mask_bg (Camera.py:515)"""
    gray = rgbs_gt.mean(dim=-1)
    return (gray < 0.95) & (gray > 0.05)


def _mask(m, n_rays, rgbs_gt=None):
    """a per-ray mask as uint8 [n_rays]; None = every ray; "gt" = mask_bg of the ground-truth colours: kept as the string for
    the fused loss head (evaluated in its kernels), formed here when `rgbs_gt` is given (the two-call form)"""
    if m is None:
        return None
    if isinstance(m, str):
        if m != "gt":
            raise ValueError(f"ls2fm.losses: unknown mask {m!r}")
        if rgbs_gt is None:
            return m
        m = bg_mask(rgbs_gt)
    m = m.reshape(-1)
    assert m.numel() == n_rays, "ray masks have one entry per ray"
    return (m if m.dtype == torch.uint8 else m.to(torch.uint8)).contiguous()


class _LossHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, normals, depth, depth_ref, rgb_gt, mask_eik, mask_dc, mask_mse, weights, ws, global_counts):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        _lib.require_device(rgb, normals, depth, depth_ref, rgb_gt, weights)
        n_samples = normals.shape[-2]
        n_rays = normals.numel() // (3 * n_samples)
        rgb_c, nrm_c, dep_c, gt_c = _lib.cf(rgb), _lib.cf(normals), _lib.cf(depth), _lib.cf(rgb_gt)
        ref_c = _lib.cf(depth_ref) if depth_ref is not None else None
        assert rgb_c.numel() == 3 * n_rays and gt_c.numel() == 3 * n_rays and dep_c.numel() == n_rays
        terms = torch.empty(8, device=rgb.device, dtype=torch.float32)      # [5] = copy of the total (own output), [6] = PSNR
        sums = torch.empty(8, device=rgb.device, dtype=torch.float64)
        _lib.check(lib.ls2fm_loss_head_fwd(_lib.ptr(rgb_c), _lib.ptr(gt_c), _lib.ptr(nrm_c), _lib.ptr(dep_c), _lib.ptr(ref_c),
                                           _lib.ptr(mask_eik), _lib.ptr(mask_dc), _lib.ptr(mask_mse), n_rays, n_samples,
                                           _lib.ptr(weights), _lib.ptr(terms), _lib.ptr(sums), _lib.ptr(ws), _lib.stream_ptr()),
                   "ls2fm_loss_head_fwd")
        from . import dist as _dist
        if _dist.is_distributed():       # sharded rays: the backward must divide by the GLOBAL counts (means over all ranks' rays)
            _dist.globalize_loss_sums(sums, global_counts)
            _lib.check(lib.ls2fm_loss_terms_from_sums(_lib.ptr(sums), _lib.ptr(weights), _lib.ptr(terms), _lib.stream_ptr()),
                       "ls2fm_loss_terms_from_sums")
        ctx.saved = (rgb_c, nrm_c, dep_c, ref_c, gt_c, mask_eik, mask_dc, mask_mse, weights, sums)
        _LAST_SUMS[rgb.device.index] = sums
        ctx.shapes = (rgb.shape, normals.shape, depth.shape, None if depth_ref is None else depth_ref.shape)
        ctx.n = (n_rays, n_samples)
        return terms[:5], terms[5]

    @staticmethod
    def backward(ctx, g, g_total):
        lib = _lib.load()
        if g is None and g_total is None:
            return (None,) * 11
        rgb_c, nrm_c, dep_c, ref_c, gt_c, mask_eik, mask_dc, mask_mse, weights, sums = ctx.saved
        n_rays, n_samples = ctx.n
        g = None if g is None else _lib.cf(g)
        g_total = None if g_total is None else _lib.cf(g_total)
        d_rgb, d_nrm, d_dep = torch.empty_like(rgb_c), torch.empty_like(nrm_c), torch.empty_like(dep_c)
        d_ref = torch.empty_like(ref_c) if (ref_c is not None and ctx.needs_input_grad[3]) else None
        _lib.check(lib.ls2fm_loss_head_bwd(_lib.ptr(rgb_c), _lib.ptr(gt_c), _lib.ptr(nrm_c), _lib.ptr(dep_c), _lib.ptr(ref_c),
                                           _lib.ptr(mask_eik), _lib.ptr(mask_dc), _lib.ptr(mask_mse), n_rays, n_samples,
                                           _lib.ptr(weights), _lib.ptr(g), _lib.ptr(g_total), _lib.ptr(d_rgb), _lib.ptr(d_nrm), _lib.ptr(d_dep),
                                           _lib.ptr(d_ref), _lib.ptr(sums), _lib.stream_ptr()), "ls2fm_loss_head_bwd")
        s_rgb, s_nrm, s_dep, s_ref = ctx.shapes
        return (d_rgb.view(s_rgb), d_nrm.view(s_nrm), d_dep.view(s_dep), None if d_ref is None else d_ref.view(s_ref),
                None, None, None, None, None, None, None)


class RenderLossHead:
    """``head = RenderLossHead(device, w_rgb=3, w_eikonal=2, w_dc=0)`` (the ``opt.loss_weight`` exponents; None = term
    off) then ``loss = head(ret, rgbs_gt, d_points=None, mask_finish=None, mask_eik=None, mask_bg=None)`` returns a
    dict with the reference's keys ``rgb_loss, eikonal_loss, DC_loss, mse, all`` (0-dim tensors; ``all`` carries the
    gradient to ``ret['rgb'], ret['normals'], ret['depth_mlp']`` and ``d_points``)."""

    def __init__(self, device, w_rgb=3.0, w_eikonal=2.0, w_dc=0.0, global_counts="allreduce"):
        w = [0.0 if x is None else 10.0 ** float(x) for x in (w_rgb, w_eikonal, w_dc)]
        self.weights = torch.tensor(w, device=device, dtype=torch.float32)
        self.global_counts = global_counts      # multi-GPU runs: see spec()

    def terms(self, ret, rgbs_gt, d_points=None, mask_finish=None, mask_eik=None, mask_bg=None):
        """-> (terms [5] = rgb, eikonal, DC, mse, all ; all as its own 0-dim tensor: backward through it costs no
        select/zero-fill kernels)"""
        normals = ret["normals"]
        n_rays = normals.numel() // (3 * normals.shape[-2])
        return _LossHead.apply(ret["rgb"], normals, ret["depth_mlp"], d_points, rgbs_gt, _mask(mask_eik, n_rays, rgbs_gt),
                               _mask(mask_finish, n_rays), _mask(mask_bg, n_rays, rgbs_gt), self.weights,
                               _workspace(normals.device), self.global_counts)

    def __call__(self, ret, rgbs_gt, **kw):
        t, total = self.terms(ret, rgbs_gt, **kw)
        return self.as_dict(t, total)

    @staticmethod
    def as_dict(t, total):
        return {"rgb_loss": t[0], "eikonal_loss": t[1], "DC_loss": t[2], "mse": t[3], "all": total}

    def spec(self, rgbs_gt, mask_finish=None, mask_eik=None, mask_bg=None, n_rays=None, global_counts=None):
        """the same loss head as a FusedLoss for `Renderer.forward_with_loss` / `ls2fm.fused.render(loss=...)`.
        global_counts (multi-GPU runs only): "allreduce" = the (sum, count) pairs are all-reduced between forward and backward,
        every rank sees the global means; "uniform" = no collective: every rank holds the same number of rays and no masks, the
        counts are scaled by the world size and a rank's terms are its SHARE of the global means (they add up over the ranks)."""
        from .fused import FusedLoss
        n_rays = rgbs_gt.numel() // 3 if n_rays is None else n_rays
        gt = _lib.cf(rgbs_gt.detach()).reshape(-1)
        _lib.require_device(gt)
        return FusedLoss(self.weights, gt, _mask(mask_eik, n_rays), _mask(mask_finish, n_rays), _mask(mask_bg, n_rays),
                         self.global_counts if global_counts is None else global_counts)

    @staticmethod
    def sums(device):
        """fp64 [8] sums and counts of the last call on this device (for global normalisation under sharding):
        S|rgb-gt|, n, S| |n|-1 |, n, S smooth_l1, n, S (rgb-gt)^2, n"""
        d = torch.device(device)
        return _LAST_SUMS[d.index if d.index is not None else torch.cuda.current_device()]


def psnr(mse):
    return -10.0 * torch.log10(mse)
