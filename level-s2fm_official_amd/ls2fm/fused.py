"""Fused-kernel entry points (filled in as the kernels land).  Until then every predicate says "no" and
the general autograd-composed form (HIP hash-grid op + torch dense layers) serves all calls."""
from __future__ import annotations


def available(field, probe) -> bool:
    return False


def can_eval_without_graph(sdf_field, xyz) -> bool:
    return False


def sdf_eval(sdf_field, xyz, want_feat=False):
    raise RuntimeError("fused sdf_eval not built")


def sphere_trace(sdf_field, o, d):
    raise RuntimeError("fused sphere_trace not built")


def can_render(renderer, opt, center, ray, sdf_field, rad_field) -> bool:
    return False


def render(renderer, opt, center, ray, sdf_field, rad_field):
    raise RuntimeError("fused render not built")
