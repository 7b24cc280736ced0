"""Fused loss head (csrc/loss_head.hip through the C ABI) against the torch restatement of the reference's expressions
(oracle/losses.py): values and gradients, with and without masks, empty masks, ragged sizes."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-6


def _inputs(n_rays, n_samples, seed, with_ref=True):
    g = torch.Generator().manual_seed(seed)
    ret = {"rgb": torch.rand(1, n_rays, 3, generator=g), "normals": torch.randn(1, n_rays, n_samples, 3, generator=g) * 1.3,
           "depth_mlp": torch.rand(1, n_rays, 1, generator=g) * 4}
    gt = torch.rand(1, n_rays, 3, generator=g)
    d_points = torch.rand(1, n_rays, generator=g) * 5 if with_ref else None     # |d| spans both smooth-L1 branches
    ret["normals"][0, 0, 0] = 0.0                                                # ||n|| = 0: gradient defined as 0
    return ret, gt, d_points


def _run(n_rays, n_samples, seed, masks, weights, with_ref=True):
    from ls2fm.losses import RenderLossHead
    from oracle.losses import loss_head
    ret, gt, d_points = _inputs(n_rays, n_samples, seed, with_ref)
    g = torch.Generator().manual_seed(seed + 1)
    mk = {k: (torch.rand(1, n_rays, generator=g) < p) if p is not None else None for k, p in masks.items()}
    # oracle (CPU, fp64 for a clean reference)
    ro = {k: v.double().requires_grad_(True) for k, v in ret.items()}
    dpo = d_points.double().requires_grad_(True) if with_ref else None
    lo = loss_head(ro, gt.double(), dpo, mk["finish"], mk["eik"], mk["bg"], *weights)
    lo["all"].backward()
    # product
    rp = {k: v.to(DEV).requires_grad_(True) for k, v in ret.items()}
    dpp = d_points.to(DEV).requires_grad_(True) if with_ref else None
    head = RenderLossHead(DEV, *weights)
    lp = head(rp, gt.to(DEV), d_points=dpp, mask_finish=None if mk["finish"] is None else mk["finish"].to(DEV),
              mask_eik=None if mk["eik"] is None else mk["eik"].to(DEV), mask_bg=None if mk["bg"] is None else mk["bg"].to(DEV))
    lp["all"].backward()
    for k in ("rgb_loss", "eikonal_loss", "DC_loss", "mse", "all"):
        a, b = lp[k].detach().cpu().double(), torch.as_tensor(lo[k]).detach().double()
        if torch.isnan(b):
            assert torch.isnan(a), k
        else:
            assert abs(float(a - b)) <= TOL * max(1.0, abs(float(b))), (k, float(a), float(b))
    for k in rp:
        want = ro[k].grad.float() if ro[k].grad is not None else torch.zeros_like(ret[k])     # term off: exact zeros
        assert rel_err(rp[k].grad.cpu(), want) < 1e-5, k
    if with_ref and dpo.grad is not None:
        assert rel_err(dpp.grad.cpu(), dpo.grad.float()) < 1e-5
    return head


@pytest.mark.parametrize("n_rays,n_samples", [(1, 1), (37, 5), (1024, 128), (513, 300)])
def test_loss_head_no_masks(n_rays, n_samples):
    _run(n_rays, n_samples, 3, dict(finish=None, eik=None, bg=None), (3.0, 2.0, 0.5))


def test_loss_head_masks():
    _run(700, 33, 4, dict(finish=0.4, eik=0.7, bg=0.8), (3.0, 2.0, 1.0))


def test_loss_head_empty_finish_mask_gives_zero_dc():
    _run(64, 8, 5, dict(finish=0.0, eik=None, bg=None), (3.0, 2.0, 1.0))


def test_loss_head_without_depth_reference_and_off_terms():
    _run(64, 8, 6, dict(finish=None, eik=None, bg=None), (3.0, None, None), with_ref=False)


def test_loss_head_is_deterministic_and_exposes_sums():
    from ls2fm.losses import RenderLossHead
    ret, gt, d_points = _inputs(1024, 128, 9)
    rp = {k: v.to(DEV) for k, v in ret.items()}
    head = RenderLossHead(DEV)
    a = head.terms(rp, gt.to(DEV), d_points=d_points.to(DEV))[0].clone()
    sums = RenderLossHead.sums(DEV).clone()
    for _ in range(3):
        assert torch.equal(a, head.terms(rp, gt.to(DEV), d_points=d_points.to(DEV))[0])
    assert float(sums[1]) == 3 * 1024 and float(sums[3]) == 1024 * 128 and float(sums[5]) == 1024
    assert abs(float(sums[0] / sums[1]) - float(a[0])) < 1e-6


def test_loss_head_backward_through_individual_terms():
    """gradient routed through the per-term outputs (not the weighted total) and through both at once"""
    from ls2fm.losses import RenderLossHead
    from oracle.losses import loss_head
    ret, gt, d_points = _inputs(200, 16, 12)
    ro = {k: v.double().requires_grad_(True) for k, v in ret.items()}
    lo = loss_head(ro, gt.double(), d_points.double(), None, None, None, 3.0, 2.0, 0.5)
    (2.0 * lo["rgb_loss"] + lo["eikonal_loss"] * 0.5 + 3.0 * lo["mse"] + 0.25 * lo["all"]).backward()
    rp = {k: v.to(DEV).requires_grad_(True) for k, v in ret.items()}
    lp = RenderLossHead(DEV, 3.0, 2.0, 0.5)(rp, gt.to(DEV), d_points=d_points.to(DEV))
    (2.0 * lp["rgb_loss"] + lp["eikonal_loss"] * 0.5 + 3.0 * lp["mse"] + 0.25 * lp["all"]).backward()
    for k in rp:
        assert rel_err(rp[k].grad.cpu(), ro[k].grad.float()) < 1e-5, k
