"""Operator-level drop-in for the one `vren` op the reference actually calls
(utils/custom_functions.py:10-31 -> Renderer.py:178, SDF.py:120).  The four other wrappers in the
reference's file (RaySphereIntersector, RayMarcher, VolumeRenderer, TruncExp) have no caller
 and are not part of the path."""
from ..ops import ray_aabb_intersect


class RayAABBIntersector:
    """`RayAABBIntersector.apply(rays_o, rays_d, center, half_size, max_hits)` ->
    [hits_cnt (N,), hits_t (N, max_hits, 2), hits_voxel_idx (N, max_hits)].

    Like the reference's wrapper (whose forward returns the extension's python list) the result is not
    tracked by autograd: near/far carry no gradient to the rays."""

    @staticmethod
    def apply(rays_o, rays_d, center, half_size, max_hits):
        return ray_aabb_intersect(rays_o.float(), rays_d.float(), center.float(), half_size.float(), max_hits)
