"""Camera-side arithmetic the stage loops need around the path (the reference's utils/camera.py).

`get_3D_points_from_depth` (camera.py:262-266) sits on the render path itself.  The rest -- pinhole projection, the pose
algebra and the ray pick of `get_center_and_ray` (camera.py:230-252) -- is what `ls2fm.stage`'s loop drivers use to turn poses,
intrinsics and a ray permutation into the `[B,R,3]` centers / rays the fused render takes, and what a bundle-adjustment step
differentiates through for its re-projection term (BA.py:124-131).  Plain torch on whatever device the inputs live on (GPU in
the loops: nothing here synchronises, so a step containing it can be captured).  Same function names and argument meaning as
the reference's module so that a caller can switch imports; written from the textbook formulas:

    world -> camera   x_c = R x_w + t                      pose = [R | t]  (3 x 4, world-to-camera)
    camera -> image   u   = K x_c                          (homogeneous; divide by u_z for pixels)
    se(3) -> SE(3)    R = I + A W + B W^2,  t = (I + B W + C W^2) u,   W = [w]_x,  theta = |w|,
                      A = sin(theta) / theta,  B = (1 - cos theta) / theta^2,  C = (theta - sin theta) / theta^3
"""
from __future__ import annotations

import math

import torch


def get_3D_points_from_depth(opt, center, ray, depth, multi_samples=False):
    """p = c + d * t for [B,R,3] rays and [B,R,N,1] depths (ray directions are NOT normalised)."""
    if multi_samples:
        center, ray = center[:, :, None], ray[:, :, None]
    return center + ray * depth


# ------------------------------------------------------------------------------------------------ projective helpers
def to_hom(X):
    """[...,k] -> [...,k+1] with a trailing 1"""
    return torch.cat([X, torch.ones_like(X[..., :1])], dim=-1)


def invert_pose(pose):
    """[...,3,4] world-to-camera -> camera-to-world: [R^T | -R^T t]"""
    R, t = pose[..., :3], pose[..., 3:]
    Rt = R.transpose(-1, -2)
    return torch.cat([Rt, -Rt @ t], dim=-1)


def world2cam(X, pose):
    """X [B,N,3] world points, pose [B,3,4] -> camera coordinates [B,N,3]"""
    return to_hom(X) @ pose.transpose(-1, -2)


def cam2world(X, pose):
    return to_hom(X) @ invert_pose(pose).transpose(-1, -2)


def cam2img(X, cam_intr):
    return X @ cam_intr.transpose(-1, -2)


def _inverse_intrinsic(cam_intr):
    """K^-1, remembered ON the tensor that owns the matrix's storage (per version and view geometry): the loops call this
    every iteration with the same constant matrix, and a 3 x 3 inverse is eight launch-bound solver kernels"""
    if cam_intr.requires_grad:
        return torch.linalg.inv(cam_intr)
    owner = cam_intr._base if cam_intr._base is not None else cam_intr
    key = (cam_intr._version, tuple(cam_intr.shape), tuple(cam_intr.stride()), cam_intr.storage_offset())
    memo = getattr(owner, "_ls2fm_inverse", None)
    if memo is None or memo[0] != key:
        memo = (key, torch.linalg.inv(cam_intr))
        try:
            owner._ls2fm_inverse = memo
        except Exception:
            pass
    return memo[1]


def img2cam(X, cam_intr):
    return X @ _inverse_intrinsic(cam_intr).transpose(-1, -2)


def mesh_grid(opt=None, H=None, W=None, device=None):
    """pixel centres [H*W, 2] as (x + 0.5, y + 0.5), row-major (camera.py:253-261)"""
    H = int(opt.H) if H is None else int(H)
    W = int(opt.W) if W is None else int(W)
    device = (opt.device if opt is not None else "cpu") if device is None else device
    ys = torch.arange(H, dtype=torch.float32, device=device) + 0.5
    xs = torch.arange(W, dtype=torch.float32, device=device) + 0.5
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([X, Y], dim=-1).view(-1, 2)


def get_center_and_ray(opt, pose, intr=None, rays_idx=None, xy_grid=None):
    """pose [B,3,4], intr [1|B,3,3] -> camera centers and (unnormalised) ray directions of the picked pixels, [B,R,3] each
    (camera.py:230-252: the SAME pixels `rays_idx` for every view)"""
    with torch.no_grad():
        grid = mesh_grid(opt, device=pose.device) if xy_grid is None else xy_grid
        if rays_idx is not None:
            grid = grid[rays_idx, :]
    B = pose.shape[0]
    grid = grid.unsqueeze(0).expand(B, -1, -1)
    in_cam = img2cam(to_hom(grid), intr)                 # points on the z = 1 plane of each camera
    origin = torch.zeros_like(in_cam)
    center = cam2world(origin, pose)
    return center, cam2world(in_cam, pose) - center


def host_inverse_intrinsic(cam_intr):
    """K^-1 (as `img2cam` uses it: computed on the matrix's own device) as 9 host floats for `camera_rays` -- call it where a host
    read-back is allowed (a loop's constructor), not inside a captured step"""
    import ctypes
    k = _inverse_intrinsic(cam_intr.reshape(3, 3))
    return (ctypes.c_float * 9)(*[float(v) for v in k.detach().reshape(-1).cpu().tolist()])


def camera_rays(kinv_host, poses=None, se3=None, xy=None, pix=None, width=0, view_sel=None, out=None, poses_out=None):
    """`get_center_and_ray` / `keypoint_rays` (and, with `se3` [V,6], `Lie.se3_to_SE3` in front) as ONE launch, without a graph
    (ls2fm_camera_rays): the pose algebra of a loop iteration's ray pick was ~95 launch-bound torch kernels.
    poses [V,3,4] or se3 [V,6]; pixels: xy [n,2] / [V,n,2] float coordinates or pix [n] long indices of a width-`width` image;
    view_sel: device long [1] -> only that view (outputs [1,n,3]).  out: (centers, rays) buffers to fill in place.
    -> (centers, rays) [V or 1, n, 3]"""
    from .. import _lib
    lib = _lib.load()
    src = poses if poses is not None else se3
    _lib.require_device(src)
    n_views = src.shape[0]
    per_view = xy is not None and xy.dim() == 3
    n = (pix.shape[0] if pix is not None else xy.shape[-2])
    v_out = 1 if view_sel is not None else n_views
    if out is None:
        out = (torch.empty(v_out, n, 3, device=src.device), torch.empty(v_out, n, 3, device=src.device))
    centers, rays = out
    if centers.numel() != v_out * n * 3 or rays.numel() != v_out * n * 3 or not (centers.is_contiguous() and rays.is_contiguous()):
        raise RuntimeError("ls2fm.utils.camera.camera_rays: out buffers must be contiguous [views, n, 3]")
    f = lambda t: None if t is None else (t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous())
    poses_c, se3_c, xy_c = f(None if poses is None else poses.detach()), f(None if se3 is None else se3.detach()), f(xy)
    pix_c = None if pix is None else (pix if (pix.dtype == torch.int64 and pix.is_contiguous()) else pix.long().contiguous())
    _lib.check(lib.ls2fm_camera_rays(_lib.ptr(poses_c), _lib.ptr(se3_c), kinv_host, _lib.ptr(xy_c), _lib.ptr(pix_c), int(width),
                                     1 if per_view else 0, _lib.ptr(view_sel), n_views, n, _lib.ptr(centers), _lib.ptr(rays),
                                     _lib.ptr(poses_out), _lib.stream_ptr()), "ls2fm_camera_rays")
    return centers, rays


# ------------------------------------------------------------------------------------------------ se(3)
_N_TERMS = 11
_SERIES = {}


def _series_constants(device, dtype):
    """exponents 2i [11], signs (-1)^i [11] and the denominators [3, 11] of the Taylor series of  A = sin(t) / t  (d_i = (2i + 1)!),
    B = (1 - cos t) / t^2  ((2i + 2)!),  C = (t - sin t) / t^3  ((2i + 3)!), the denominators built by the running products the
    reference's loops use (11 terms: exact to fp32 for |theta| < pi, and smooth through theta = 0 where the closed forms are 0 / 0)"""
    key = (str(device), dtype)
    if key not in _SERIES:
        rows = []
        for first, step in ((1.0, lambda i: (2 * i) * (2 * i + 1)), (2.0, lambda i: (2 * i + 1) * (2 * i + 2)),
                            (6.0, lambda i: (2 * i + 2) * (2 * i + 3))):
            denom, row = float(first), []
            for i in range(_N_TERMS):
                if i > 0:
                    denom *= step(i)
                row.append(denom)
            rows.append(row)
        _SERIES[key] = (torch.arange(_N_TERMS, device=device, dtype=dtype) * 2,
                        torch.tensor([(-1.0) ** i for i in range(_N_TERMS)], device=device, dtype=dtype),
                        torch.tensor(rows, dtype=torch.float64).to(device=device, dtype=dtype))
    return _SERIES[key]


def _abc(theta):
    """theta [..., 1, 1] -> A, B, C of the same shape.  The reference evaluates each series term by term,
    total += (-1)^i theta^(2i) / d_i; as a Python loop that is ~130 launch-bound elementwise kernels per call and as many again
    in its backward -- a captured BA iteration spent more launches here than in the render.  Same terms, same left-to-right
    order of the sum (the unstable trajectories of the loop goldens notice a re-associated sum), but the powers and quotients
    of all three series are formed at once: 13 kernels."""
    expo, sign, denom = _series_constants(theta.device, theta.dtype)
    terms = (sign * theta.unsqueeze(-1) ** expo).unsqueeze(-2) / denom                    # [..., 1, 1, 3, 11]
    total = terms[..., 0]
    for i in range(1, _N_TERMS):
        total = total + terms[..., i]
    return total[..., 0], total[..., 1], total[..., 2]


def skew(w):
    w0, w1, w2 = w.unbind(dim=-1)
    zero = torch.zeros_like(w0)
    return torch.stack([torch.stack([zero, -w2, w1], dim=-1), torch.stack([w2, zero, -w0], dim=-1),
                        torch.stack([-w1, w0, zero], dim=-1)], dim=-2)


class Lie:
    """SO(3) / SE(3) exponential and logarithm maps (camera.py:63-147)"""

    skew_symmetric = staticmethod(skew)

    @staticmethod
    def so3_to_SO3(w):
        W = skew(w)
        A, B, _ = _abc(w.norm(dim=-1)[..., None, None])
        return torch.eye(3, device=w.device, dtype=w.dtype) + A * W + B * (W @ W)

    @staticmethod
    def se3_to_SE3(wu):
        w, u = wu[..., :3], wu[..., 3:]
        W = skew(w)
        W2 = W @ W
        A, B, C = _abc(w.norm(dim=-1)[..., None, None])
        eye = torch.eye(3, device=w.device, dtype=w.dtype)
        R = eye + A * W + B * W2
        V = eye + B * W + C * W2
        return torch.cat([R, V @ u[..., None]], dim=-1)

    @staticmethod
    def SO3_to_so3(R, eps=1e-7):
        cos = ((R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2] - 1) / 2).clamp(-1 + eps, 1 - eps)
        theta = torch.remainder(torch.acos(cos), math.pi)[..., None, None]
        A, _, _ = _abc(theta)
        log = (R - R.transpose(-2, -1)) / (2 * A + 1e-8)
        return torch.stack([log[..., 2, 1], log[..., 0, 2], log[..., 1, 0]], dim=-1)

    @staticmethod
    def SE3_to_se3(Rt, eps=1e-8):
        R, t = Rt[..., :3], Rt[..., 3:]
        w = Lie.SO3_to_so3(R)
        W = skew(w)
        theta = w.norm(dim=-1)[..., None, None]
        A, B, _ = _abc(theta)
        inv_v = torch.eye(3, device=w.device, dtype=w.dtype) - 0.5 * W + (1 - A / (2 * B)) / (theta ** 2 + eps) * (W @ W)
        return torch.cat([w, (inv_v @ t)[..., 0]], dim=-1)


class _Se3Exp(torch.autograd.Function):
    """ls2fm_se3_exp_fwd / _bwd: `Lie.se3_to_SE3` with its gradient as one launch each way (GPU tensors)"""

    @staticmethod
    def forward(ctx, wu):
        from .. import _lib
        lib = _lib.load()
        x = wu.detach().float().contiguous()
        _lib.require_device(x)
        n = x.numel() // 6
        out = torch.empty(*x.shape[:-1], 3, 4, device=x.device)
        _lib.check(lib.ls2fm_se3_exp_fwd(_lib.ptr(x), n, _lib.ptr(out), _lib.stream_ptr()), "ls2fm_se3_exp_fwd")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import _lib
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        g = g.detach().float().contiguous()
        d = torch.empty_like(x)
        _lib.check(lib.ls2fm_se3_exp_bwd(_lib.ptr(x), _lib.ptr(g), x.numel() // 6, _lib.ptr(d), _lib.stream_ptr()), "ls2fm_se3_exp_bwd")
        return d


def se3_to_SE3_fused(wu):
    """`Lie.se3_to_SE3` through the fused kernels when `wu` lives on the GPU (differentiable once), else the torch expression"""
    return _Se3Exp.apply(wu) if wu.is_cuda else Lie.se3_to_SE3(wu)


lie = Lie()
