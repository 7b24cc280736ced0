"""hipGraph capture of a whole step (ls2fm.graph.CapturedStep): replays reproduce the eager step (to the last-bit
freedom of the few float atomics: split coarse levels, d beta) and pick up in-place changes of the inputs."""
import pytest
import torch

from conftest import rel_err
from helpers import named_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_captured_step_matches_eager_and_tracks_inputs():
    from ls2fm.graph import CapturedStep
    from ls2fm.losses import RenderLossHead
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
    opt = make_options("BlendedMVS", device=DEV, dual_field=True, sample_intvs=32,
                       hash_encoding=dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16))
    torch.manual_seed(5)
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for mod in (sdf, rad):
            for name, p in mod.named_parameters():
                if name.endswith("embedder_obj.params"):
                    p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * 0.1).to(DEV))
                if name.endswith("mlp.0.weight_v") and "Rad_dec" not in name:
                    p[:, 3:] = (torch.randn(p[:, 3:].shape, generator=g) * 0.05).to(DEV)
    params = list(sdf.parameters()) + list(rad.parameters())
    center = torch.tensor([0.0, 0.0, -4.0]).repeat(1, 96, 1).to(DEV)
    ray = (torch.tensor([0.0, 0.0, 1.0]).repeat(1, 96, 1) + 0.15 * torch.randn(1, 96, 3, generator=g)).to(DEV)
    gt = torch.rand(1, 96, 3, generator=g).to(DEV)
    head = RenderLossHead(DEV, 3.0, 2.0, None)

    def step():
        for p in params:
            p.grad = None
        ret = ren.forward(opt, center, ray, sdf, rad)
        loss = head.terms(ret, gt)[1]
        loss.backward()
        return loss

    def snapshot():
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in {**named_grads(sdf), **{"r." + k: v for k, v in named_grads(rad).items()}}.items()}

    cap = CapturedStep(step, params)  # capture first: autograd's grad-accumulator nodes must not be tied to the legacy
    replays = []                      # default stream by an earlier eager backward (torch would sync with it mid-capture)
    for _ in range(2):
        loss_g = cap.replay().clone()
        replays.append((loss_g, snapshot()))
    loss_e = step().clone()
    eager = snapshot()
    for loss_g, got in replays:
        assert rel_err(loss_g, loss_e) < 1e-6
        for k in eager:
            assert rel_err(got[k], eager[k]) < 1e-6, k
    # new rays written in place are picked up by the next replay
    ray2 = (torch.tensor([0.0, 0.0, 1.0]).repeat(1, 96, 1) + 0.1 * torch.randn(1, 96, 3, generator=g)).to(DEV)
    ray.copy_(ray2)
    loss_g2 = cap.replay().clone()
    got2 = snapshot()
    loss_e2 = step().clone()
    eager2 = snapshot()
    assert rel_err(loss_g2, loss_e2) < 1e-6 and rel_err(loss_g2, loss_e) > 1e-4
    for k in eager2:
        assert rel_err(got2[k], eager2[k]) < 1e-6, k


def test_captured_step_with_interleaved_tables_follows_parameter_updates(monkeypatch):
    """LS2FM_DUAL_TABLE=version inside a captured step: the host cannot re-check parameter versions at replay, so the
    table interleave is captured with the step and every replay renders the CURRENT tables"""
    from ls2fm import fused
    from ls2fm.graph import CapturedStep
    from ls2fm.optim import FusedAdam
    from ls2fm.losses import RenderLossHead
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
    monkeypatch.setattr(fused, "_DUAL_TABLE", "version")
    opt = make_options("DTU", device=DEV, dual_field=True, sample_intvs=32,
                       hash_encoding=dict(n_levels=6, n_features_per_level=2, log2_hashmap_size=12, base_resolution=16))
    torch.manual_seed(9)
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for mod in (sdf, rad):
            for name, p in mod.named_parameters():
                if name.endswith("embedder_obj.params"):
                    p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * 0.1).to(DEV))
                if name.endswith("mlp.0.weight_v") and "Rad_dec" not in name:
                    p[:, 3:] = (torch.randn(p[:, 3:].shape, generator=g) * 0.05).to(DEV)
    params = list(sdf.parameters()) + list(rad.parameters())
    center = torch.tensor([0.0, 0.0, -2.5]).repeat(1, 64, 1).to(DEV)
    ray = (torch.tensor([0.0, 0.0, 1.0]).repeat(1, 64, 1) + 0.15 * torch.randn(1, 64, 3, generator=g)).to(DEV)
    gt = torch.rand(1, 64, 3, generator=g).to(DEV)
    head = RenderLossHead(DEV, 3.0, 2.0, None)

    def step():
        for p in params:
            p.grad = None
        loss = head.terms(ren.forward(opt, center, ray, sdf, rad), gt)[1]
        loss.backward()
        return loss

    cap = CapturedStep(step, params)
    optim = FusedAdam(params, lr=5e-3)
    losses_g = []
    for _ in range(3):
        losses_g.append(cap.replay().clone())
        optim.step()                                  # moves both tables between replays
    torch.cuda.synchronize()
    assert rel_err(losses_g[1], losses_g[0]) > 1e-5   # the updates are visible to the replays
    # same trajectory without graph and without the interleaved copy
    monkeypatch.setattr(fused, "_DUAL_TABLE", "off")
    loss_now = step().clone()                          # state after 3 updates
    loss_replay = cap.replay().clone()
    assert rel_err(loss_replay, loss_now) < 1e-6
