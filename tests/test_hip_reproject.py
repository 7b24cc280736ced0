"""ls2fm.stage.reprojection_term (ls2fm_reproject_fwd / _bwd: the per-observation block of a BA iteration, pipelines/BA.py:126-147,
199-202) against the same term written with torch ops the way the reference writes it -- value, the counted mask, and the
gradients w.r.t. the points and the poses; incl. observations masked by the SDF bound, a view without observations and the
all-masked case (the term and its gradients are 0)."""
import pytest
import torch

from ls2fm import stage
from ls2fm.utils import camera as cam

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _torch_term(x, poses, obs_view, K, obs_uv, sdf, bound):
    in_cam = cam.world2cam(x.unsqueeze(1), poses[obs_view])
    uv = cam.cam2img(in_cam, K.expand(in_cam.shape[0], 3, 3))
    uv = (uv / (uv[..., 2:] + 1e-6))[..., :2].squeeze(1)
    on = (sdf.abs() < bound) & ~torch.isinf(uv).any(dim=-1)
    err = torch.where(on, (uv - obs_uv).norm(dim=-1), torch.zeros((), device=uv.device))
    n = on.sum()
    robust = torch.where(on, 2 * torch.log(1 + err ** 2 / 4), torch.zeros((), device=uv.device))
    return torch.where(n > 0, 0.5 * robust.sum() / n.clamp_min(1) + 0.5 * err.sum() / n.clamp_min(1), torch.zeros((), device=uv.device)), on


@pytest.mark.parametrize("case", ["mixed", "empty_view", "all_masked"])
def test_reprojection_term_vs_torch(case):
    gen = torch.Generator().manual_seed(5)
    counts = [300, 0, 513] if case == "empty_view" else [257, 64, 400]
    V, n = len(counts), sum(counts)
    se3 = torch.cat([0.2 * torch.randn(V, 3, generator=gen), torch.randn(V, 3, generator=gen) * 0.3 + torch.tensor([0.0, 0.0, 4.0])], dim=1)
    poses = cam.lie.se3_to_SE3(se3).to(DEV).requires_grad_(True)
    x = (torch.randn(n, 3, generator=gen) * 0.5).to(DEV).requires_grad_(True)
    K = torch.tensor([[60.0, 0.0, 32.0], [0.0, 60.0, 24.0], [0.0, 0.0, 1.0]], device=DEV)
    obs_view = torch.cat([torch.full((c,), v, dtype=torch.long) for v, c in enumerate(counts)]).to(DEV)
    obs_uv = (torch.rand(n, 2, generator=gen) * torch.tensor([64.0, 48.0])).to(DEV)
    sdf = (torch.randn(n, generator=gen) * 0.02).to(DEV)
    bound = 0.0 if case == "all_masked" else 0.03
    view_start = torch.tensor([0] + counts, dtype=torch.int32).cumsum(0).to(torch.int32).to(DEV)
    got, counted = stage.reprojection_term(x, poses, view_start, stage.host_intrinsic(K), obs_uv, sdf, bound)
    (3.0 * got).backward()
    gx, gp = x.grad.clone(), poses.grad.clone()
    x.grad = poses.grad = None
    ref, on = _torch_term(x, poses, obs_view, K, obs_uv, sdf, bound)
    (3.0 * ref).backward()
    assert torch.equal(counted, on)
    assert abs(float(got) - float(ref)) <= 2e-6 * max(abs(float(ref)), 1e-6), (float(got), float(ref))
    if case == "all_masked":
        assert float(got) == 0.0 and float(gx.abs().max()) == 0.0 and float(gp.abs().max()) == 0.0
        return
    assert float((gx - x.grad).abs().max()) <= 2e-5 * float(x.grad.abs().max())
    assert float((gp - poses.grad).abs().max()) <= 2e-5 * float(poses.grad.abs().max())
