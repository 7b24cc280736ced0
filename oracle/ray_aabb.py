"""Oracle: ray / axis-aligned-box slab test (``vren.ray_aabb_intersect``).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference:
``vren`` is built from kwea123/ngp_pl ``models/csrc`` (unpinned git HEAD, wheel 2.0
per env.yaml:251) and is not in /root/reference.  Call sites that fix the observable
contract: utils/custom_functions.py:10-31 (docstring: hits_t is -1 where nothing is
hit, sorted near->far), models/Renderer.py:178-180 and models/SDF.py:120-122 (one
box, ``max_hits = 1``; ``hits_t[:, 0, 0]`` is used as near, ``hits_t[:, 0, 1]`` as far).

Published algorithm (ngp_pl ``intersection.cu``), SURVEY.md Appendix A.1:
``inv = 1/d`` on the UNNORMALISED direction; per axis ``t_lo = (c-h-o)*inv``,
``t_hi = (c+h-o)*inv``; ``t1 = max_axes(fmin(t_lo,t_hi))``, ``t2 = min_axes(fmax(..))``;
miss when ``t1 > t2`` or ``t2 <= 0`` -> (-1,-1); hit -> (max(t1,0), t2).
fmin/fmax have C semantics (a NaN operand is ignored), which matters for rays with a
zero direction component.

The reference's wrapper returns a python *list* so autograd never tracks the outputs
(SURVEY.md 8a row a1): near/far are constants w.r.t. the camera pose.
"""
from __future__ import annotations

import torch


def ray_aabb_intersect(rays_o: torch.Tensor, rays_d: torch.Tensor, center: torch.Tensor,
                       half_size: torch.Tensor, max_hits: int = 1):
    """rays_o, rays_d [N,3]; center, half_size [V,3] -> [hits_cnt i32[N], hits_t f32[N,max_hits,2],
    hits_voxel_idx i64[N,max_hits]] as a python list (outputs detached)."""
    with torch.no_grad():
        o = rays_o.detach().float()
        d = rays_d.detach().float()
        n = o.shape[0]
        n_vox = center.shape[0]
        hits_cnt = torch.zeros(n, dtype=torch.int32)
        hits_t = torch.full((n, max_hits, 2), -1.0, dtype=torch.float32)
        hits_idx = torch.full((n, max_hits), -1, dtype=torch.int64)
        inv = 1.0 / d
        cand_t = []
        for v in range(n_vox):
            c = center[v].detach().float()
            h = half_size[v].detach().float()
            t_lo = (c - h - o) * inv
            t_hi = (c + h - o) * inv
            a = torch.fmin(t_lo, t_hi)
            b = torch.fmax(t_lo, t_hi)
            t1 = torch.fmax(torch.fmax(a[:, 0], a[:, 1]), a[:, 2])
            t2 = torch.fmin(torch.fmin(b[:, 0], b[:, 1]), b[:, 2])
            miss = t1 > t2
            t1 = torch.where(miss, torch.full_like(t1, -1.0), t1)
            t2 = torch.where(miss, torch.full_like(t2, -1.0), t2)
            hit = t2 > 0
            cand_t.append((hit, torch.clamp_min(t1, 0.0), t2, v))
        # keep up to max_hits per ray, ordered by near t (a no-op for the single scene box)
        for r in range(n) if n_vox > 1 else ():
            hits = sorted([(float(t1[r]), float(t2[r]), v) for hit, t1, t2, v in cand_t if bool(hit[r])])
            hits_cnt[r] = len(hits)
            for k, (a_, b_, v) in enumerate(hits[:max_hits]):
                hits_t[r, k, 0], hits_t[r, k, 1], hits_idx[r, k] = a_, b_, v
        if n_vox == 1:
            hit, t1, t2, v = cand_t[0]
            hits_cnt = hit.to(torch.int32)
            hits_t[:, 0, 0] = torch.where(hit, t1, torch.full_like(t1, -1.0))
            hits_t[:, 0, 1] = torch.where(hit, t2, torch.full_like(t2, -1.0))
            hits_idx[:, 0] = torch.where(hit, torch.zeros_like(hits_idx[:, 0]), hits_idx[:, 0])
    return [hits_cnt.to(rays_o.device), hits_t.to(rays_o.device), hits_idx.to(rays_o.device)]


def near_far(rays_o: torch.Tensor, rays_d: torch.Tensor, center: torch.Tensor, half_size: torch.Tensor):
    """Convenience: (near, far) each [N] for the single scene box."""
    _, t, _ = ray_aabb_intersect(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), center.reshape(1, 3),
                                 half_size.reshape(1, 3), 1)
    return t[:, 0, 0], t[:, 0, 1]
