"""torch.autograd wrappers over the C ABI (include/ls2fm.h).  Device memory, streams and autograd
plumbing only -- all arithmetic proportional to rays x samples happens in the HIP kernels.

grid_encode            tcnn.Encoding forward/backward/double backward   (models/base.py:17,37)
ray_aabb_intersect     vren.ray_aabb_intersect                          (utils/custom_functions.py:28-31)
grid_indices           debug / parity helper (hash indices must be bit-exact)
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import cf, check, ptr, require_device, stream_ptr


# --------------------------------------------------------------------------------------------
# hash-grid encoding with first- and second-order autograd
# --------------------------------------------------------------------------------------------
class _GridEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, desc):
        require_device(x, table)
        x, table = cf(x), cf(table)
        n = x.shape[0]
        y = torch.empty(n, desc.n_levels * 2, device=x.device, dtype=torch.float32)
        check(_lib.load().ls2fm_grid_encode_fwd(ctypes.byref(desc), ptr(x), ptr(table), n, ptr(y), None, stream_ptr()),
              "ls2fm_grid_encode_fwd")
        ctx.save_for_backward(x, table)
        ctx.desc = desc
        return y

    @staticmethod
    def backward(ctx, dy):
        x, table = ctx.saved_tensors
        dx, dtable = _GridEncodeBackward.apply(x, table, dy, ctx.desc, ctx.needs_input_grad[0],
                                               ctx.needs_input_grad[1])
        return (dx if ctx.needs_input_grad[0] else None), (dtable if ctx.needs_input_grad[1] else None), None


class _GridEncodeBackward(torch.autograd.Function):
    """(x, table, dy) -> (dx, dtable); differentiable itself so `create_graph=True` works
    (the reference builds normals with autograd.grad(create_graph=True), models/SDF.py:107-113)."""

    @staticmethod
    def forward(ctx, x, table, dy, desc, want_dx, want_dtable):
        dy = cf(dy)
        n = x.shape[0]
        dx = torch.empty_like(x) if want_dx else None
        dtable = torch.zeros_like(table) if want_dtable else None
        check(_lib.load().ls2fm_grid_encode_bwd(ctypes.byref(desc), ptr(x), ptr(table), ptr(dy), n, ptr(dtable),
                                                ptr(dx), stream_ptr()), "ls2fm_grid_encode_bwd")
        ctx.save_for_backward(x, table, dy)
        ctx.desc = desc
        ctx.want = (want_dx, want_dtable)
        z = x.new_zeros(())
        return (dx if want_dx else z), (dtable if want_dtable else z)

    @staticmethod
    def backward(ctx, ddx, ddtable):
        x, table, dy = ctx.saved_tensors
        desc = ctx.desc
        want_dx, want_dtable = ctx.want
        need_x, need_table, need_dy = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        lib = _lib.load()
        n = x.shape[0]
        g_x = g_table = g_dy = None
        if want_dx and ddx is not None:
            ddx = cf(ddx)
            g_dy = torch.empty_like(dy) if need_dy else None
            g_table = torch.zeros_like(table) if need_table else None
            g_x = torch.empty_like(x) if need_x else None
            check(lib.ls2fm_grid_encode_bwd_bwd(ctypes.byref(desc), ptr(x), ptr(table), ptr(dy), ptr(ddx), n,
                                                ptr(g_dy), ptr(g_table), ptr(g_x), stream_ptr()),
                  "ls2fm_grid_encode_bwd_bwd")
        if want_dtable and ddtable is not None:
            # dtable = S(x)^T dy is linear in dy:  d/d(dy) = interpolate ddtable at x ; d/dx = dy . J(x; ddtable)
            ddtable = cf(ddtable)
            y2 = torch.empty_like(dy)
            jac = torch.empty(n, dy.shape[1], 3, device=x.device, dtype=torch.float32) if need_x else None
            check(lib.ls2fm_grid_encode_fwd(ctypes.byref(desc), ptr(x), ptr(ddtable), n, ptr(y2), ptr(jac),
                                            stream_ptr()), "ls2fm_grid_encode_fwd")
            if need_dy:
                g_dy = y2 if g_dy is None else g_dy + y2
            if need_x:
                gx2 = (jac * dy[:, :, None]).sum(dim=1)
                g_x = gx2 if g_x is None else g_x + gx2
        return g_x, g_table, g_dy, None, None, None


def grid_encode(x: torch.Tensor, table: torch.Tensor, desc: _lib.GridDesc) -> torch.Tensor:
    """x [M,3] normalised positions, table flat fp32 -> [M, L*2]."""
    if x.dim() != 2 or x.shape[1] != 3:
        raise ValueError(f"grid_encode expects x of shape [M,3], got {tuple(x.shape)}")
    return _GridEncode.apply(x, table, desc)


def grid_indices(x: torch.Tensor, desc: _lib.GridDesc) -> torch.Tensor:
    """level-local corner entry indices, int64 [M, L, 8] (values < 2^32)."""
    require_device(x)
    x = cf(x)
    out = torch.empty(x.shape[0], desc.n_levels, 8, device=x.device, dtype=torch.int32)
    check(_lib.load().ls2fm_grid_indices(ctypes.byref(desc), ptr(x), x.shape[0], ptr(out), stream_ptr()),
          "ls2fm_grid_indices")
    return out.to(torch.int64) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------
# ray / AABB
# --------------------------------------------------------------------------------------------
def ray_aabb_intersect(rays_o, rays_d, center, half_size, max_hits: int):
    """Same contract as vren.ray_aabb_intersect: returns a python *list*
    [hits_cnt i32[N], hits_t f32[N,max_hits,2], hits_voxel_idx i64[N,max_hits]] of fresh tensors that
    autograd does not track (near/far are constants w.r.t. the pose)."""
    require_device(rays_o, rays_d, center, half_size)
    with torch.no_grad():
        o, d = cf(rays_o.detach()), cf(rays_d.detach())
        c, h = cf(center.detach()).view(-1, 3), cf(half_size.detach()).view(-1, 3)
        n = o.shape[0]
        cnt = torch.empty(n, device=o.device, dtype=torch.int32)
        t = torch.empty(n, max_hits, 2, device=o.device, dtype=torch.float32)
        idx = torch.empty(n, max_hits, device=o.device, dtype=torch.int64)
        check(_lib.load().ls2fm_ray_aabb_intersect(ptr(o), ptr(d), ptr(c), ptr(h), n, c.shape[0], int(max_hits),
                                                   ptr(cnt), ptr(t), ptr(idx), stream_ptr()),
              "ls2fm_ray_aabb_intersect")
    return [cnt, t, idx]
