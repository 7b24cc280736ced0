"""ls2fm.utils.camera (the reference's utils/camera.py surface the stage loops use) against data recorded from the reference:
the ray pick `get_center_and_ray` (camera.py:230-252) against the caller-level goldens' centers / rays, the SE(3) <-> se(3)
maps (camera.py:63-147) against the poses / se(3) parameters of the stage-loop goldens.  CPU."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from ls2fm.utils import camera


@pytest.mark.parametrize("case", ["caller_dtu_dual", "caller_eth3d_single"])
def test_get_center_and_ray_matches_reference(case):
    g = load_golden(case)
    poses, intr = torch.from_numpy(g["poses"]), torch.from_numpy(g["intrinsic"])
    grid = camera.mesh_grid(H=int(g["H"]), W=int(g["W"]), device="cpu")
    centers, rays = camera.get_center_and_ray(None, poses, intr=intr.unsqueeze(0), rays_idx=torch.from_numpy(g["rays_idx"]), xy_grid=grid)
    assert np.allclose(centers.numpy(), g["centers"], rtol=1e-6, atol=1e-6)
    assert np.allclose(rays.numpy(), g["rays"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", ["stage_refine_dtu_dual", "stage_refine_eth3d_single", "stage_ba_dtu_dual"])
def test_lie_maps_match_reference(case):
    g = load_golden(case)
    poses, se3 = torch.from_numpy(g["poses"]), torch.from_numpy(g["se3"])
    assert np.allclose(camera.lie.SE3_to_se3(poses).numpy(), g["se3"], rtol=1e-5, atol=1e-6)
    back = camera.lie.se3_to_SE3(se3)
    assert np.allclose(back.numpy(), g["poses"], rtol=1e-5, atol=2e-6)
    if "se3_final" in g:                       # poses after the reference's 20 BA steps: still rigid
        R = camera.lie.se3_to_SE3(torch.from_numpy(g["se3_final"]))[..., :3]
        assert np.allclose((R @ R.transpose(-1, -2)).numpy(), np.eye(3)[None].repeat(R.shape[0], 0), atol=1e-5)


def test_projection_round_trip():
    gen = torch.Generator().manual_seed(0)
    wu = 0.3 * torch.randn(5, 6, generator=gen)
    pose = camera.lie.se3_to_SE3(wu)
    X = torch.randn(5, 7, 3, generator=gen)
    assert torch.allclose(camera.cam2world(camera.world2cam(X, pose), pose), X, atol=1e-5)
    K = torch.tensor([[30.0, 0.0, 16.0], [0.0, 30.0, 12.0], [0.0, 0.0, 1.0]])
    assert torch.allclose(camera.img2cam(camera.cam2img(X, K[None]), K[None]), X, atol=1e-4)
    assert torch.allclose(camera.lie.so3_to_SO3(torch.zeros(3)), torch.eye(3))
