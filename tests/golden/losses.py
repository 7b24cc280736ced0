"""Scalar losses shared by the golden generator and the parity tests, so that the
reference, the oracle and the HIP path are all differentiated through the very same
expression.  Mirrors the *kinds* of terms the reference's stages apply to the renderer's
outputs (rgb L1: pipelines/Camera.py:535; eikonal on ``normals``: Initialization.py:257-258;
depth consistency on ``depth_mlp``: Camera.py:522-523) plus terms on ``normal_mlp`` and
``sdfs_volume`` so every output of the path carries gradient."""
import torch


def render_loss(ret, rgb_target, nm_dir):
    eik = ((ret["normals"].norm(dim=-1) - 1.0) ** 2).mean()
    rgb = (ret["rgb"] - rgb_target).abs().mean()
    depth = torch.nn.functional.smooth_l1_loss(ret["depth_mlp"], torch.full_like(ret["depth_mlp"], 1.5))
    nm = (ret["normal_mlp"] * nm_dir).sum(dim=-1).mean()
    vol = (ret["sdfs_volume"] ** 2).mean()
    return 10.0 * rgb + 1.0 * eik + 0.5 * depth + 0.1 * nm + 0.3 * vol


def tracing_loss(d_pred, sdf_last):
    return (d_pred ** 2).mean() + 0.5 * sdf_last.abs().mean()
