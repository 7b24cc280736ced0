"""End-to-end: a short optimisation (Renderer.forward -> RenderLossHead -> backward -> FusedAdam + ExponentialLR, the
stage loop of Initialization.py:149-179 in miniature) through the fused HIP path follows the same trajectory as the
general composition with plain torch ops and torch.optim.Adam, and the loss goes down."""
import pytest
import torch

import losses
from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(seed):
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
    opt = make_options("DTU", device=DEV, dual_field=True, sample_intvs=48,
                       hash_encoding=dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=15, base_resolution=16))
    torch.manual_seed(seed)
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in (sdf, rad):
            for name, p in mod.named_parameters():
                if name.endswith("embedder_obj.params"):
                    p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05).to(DEV))
                if name.endswith("mlp.0.weight_v") and "Rad_dec" not in name:
                    p[:, 3:] = (torch.randn(p[:, 3:].shape, generator=g) * 0.05).to(DEV)
    center = torch.tensor([0.0, 0.0, -2.2]).repeat(1, 256, 1).to(DEV)
    ray = (torch.tensor([0.0, 0.0, 1.0]).repeat(1, 256, 1) + 0.2 * torch.randn(1, 256, 3, generator=g)).to(DEV)
    gt = torch.rand(1, 256, 3, generator=g).to(DEV)
    return opt, sdf, rad, ren, center, ray, gt


def test_fused_training_matches_composed_training():
    from ls2fm.losses import RenderLossHead
    from ls2fm.optim import FusedAdam
    from oracle.losses import loss_head as torch_loss_head
    opt, sdf_a, rad_a, ren, center, ray, gt = _setup(11)
    _, sdf_b, rad_b, _, _, _, _ = _setup(11)                      # identical second replica (weight_norm blocks deepcopy)
    sdf_b.load_state_dict(sdf_a.state_dict()); rad_b.load_state_dict(rad_a.state_dict())
    pa = list(sdf_a.parameters()) + list(rad_a.parameters())
    pb = list(sdf_b.parameters()) + list(rad_b.parameters())
    oa = FusedAdam(pa, lr=2e-3, betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(pb, lr=2e-3, betas=(0.9, 0.99), eps=1e-15)
    sa = torch.optim.lr_scheduler.ExponentialLR(oa, 0.98)
    sb = torch.optim.lr_scheduler.ExponentialLR(ob, 0.98)
    head = RenderLossHead(DEV, w_rgb=0.0, w_eikonal=-1.0, w_dc=None)
    la, lb = [], []
    for it in range(12):
        oa.zero_grad(set_to_none=True)
        loss = head(ren.forward(opt, center, ray, sdf_a, rad_a), gt)["all"]
        loss.backward(); oa.step(); sa.step(); la.append(float(loss.detach()))
        ob.zero_grad(set_to_none=True)
        ret = ren.forward_composed(opt, center, ray, sdf_b, rad_b)
        loss = torch_loss_head(ret, gt, w_rgb=0.0, w_eikonal=-1.0, w_dc=None)["all"]
        loss.backward(); ob.step(); sb.step(); lb.append(float(loss.detach()))
    assert la[-1] < 0.9 * la[0], la                                    # it optimises
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-3 * abs(y), (la, lb)                   # same trajectory (Adam amplifies last-bit noise)
    # parameters: Adam (eps = 1e-15) turns last-bit gradient differences of barely-touched table entries into full
    # +-lr steps, so the tables are not comparable element-wise; the dense layers are
    for (n, p), q in zip(list(sdf_a.named_parameters()) + list(rad_a.named_parameters()), pb):
        if not n.endswith("embedder_obj.params"):
            assert rel_err(p.detach().cpu(), q.detach().cpu()) < 5e-2, n
