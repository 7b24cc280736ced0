"""The one helper of the reference's utils/camera.py that sits on the path (camera.py:262-266)."""


def get_3D_points_from_depth(opt, center, ray, depth, multi_samples=False):
    """p = c + d * t for [B,R,3] rays and [B,R,N,1] depths (ray directions are NOT normalised)."""
    if multi_samples:
        center, ray = center[:, :, None], ray[:, :, None]
    return center + ray * depth
