"""SURVEY 8f-4 on the GPU: `model.ckpt` files in the reference's layout (utils/util.py:198-218 read, :239-259 write) drive the
HIP path -- a checkpoint assembled from the reference's own state dicts restores into fresh device modules whose fused render
reproduces the reference's recorded outputs and gradients, and a `RenderStage` (FusedAdam with its device-resident step count
and learning rate, eager and captured, also one whose hipGraph already exists: round-3 advisor finding) resumed from a
checkpoint continues the trajectory of the run that was never interrupted."""
from types import SimpleNamespace

import pytest
import torch

import losses
from conftest import load_golden, rel_err
from helpers import named_grads, options_for
from ls2fm import fused, stage
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from ls2fm.options import make_options
from ls2fm.utils import util
from test_hip_fused_render import _randomized, _rays

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", ["eth3d_dual", "dtu_single"])
def test_reference_checkpoint_restores_into_the_fused_render(case, manifest, tmp_path):
    """a file with the reference's keys (what utils/util.py:239-259 writes: CPU or GPU tensors under sdf_func / color_func)
    -> restore_checkpoint_sfm -> fresh SDF / RadF on the device -> fused render == the reference's recorded outputs and
    gradients; then our own save_checkpoint_sfm -> restore (resume) round trip gives the same bits"""
    g = load_golden(case)
    meta = manifest[case]
    opt = options_for(meta, DEV)
    opt.output_path = str(tmp_path)
    theirs = dict(epoch=None, iter=40,
                  sdf_func={k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sdf/")},
                  color_func={k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("rad/")},
                  cam_info=None, pts3d_info=None)
    torch.save(theirs, tmp_path / "theirs.ckpt")
    torch.manual_seed(123)                                     # fresh modules: different initial weights
    model = SimpleNamespace(sdf_func=SDF(opt).to(DEV), color_func=RadF(opt).to(DEV))
    assert util.restore_checkpoint_sfm(opt, model, load_name=str(tmp_path / "theirs.ckpt")) == (None, None)
    ren = Renderer(opt)

    def check(sdf, rad):
        center = torch.from_numpy(g["center"]).to(DEV).requires_grad_(True)
        ray = torch.from_numpy(g["ray"]).to(DEV).requires_grad_(True)
        assert fused.can_render(ren, opt, center, ray, sdf, rad)
        ret = ren.forward(opt=opt, center=center, ray=ray, SDF_Field=sdf, Rad_Field=rad)
        for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
            assert rel_err(ret[k].cpu(), g[f"ret/{k}"]) < 2e-5, k
        sdf.zero_grad(); rad.zero_grad()
        losses.render_loss(ret, torch.from_numpy(g["rgb_target"]).to(DEV), torch.from_numpy(g["nm_dir"]).to(DEV)).backward()
        for name, mod in (("sdf", sdf), ("rad", rad)):
            for k, v in named_grads(mod).items():
                if k != "beta":
                    assert rel_err(v, g[f"render_grad/{name}/{k}"]) < 1e-4, (name, k)
        return {k: v.detach().clone() for k, v in ret.items()}

    first = check(model.sdf_func, model.color_func)
    util.save_checkpoint_sfm(opt, model, ep=None, it=41)
    mine = torch.load(tmp_path / "model.ckpt", weights_only=False)
    assert set(mine) == {"epoch", "iter", "sdf_func", "color_func", "cam_info", "pts3d_info"}
    assert {k: tuple(v.shape) for k, v in mine["sdf_func"].items()} == {k: tuple(v.shape) for k, v in theirs["sdf_func"].items()}
    torch.manual_seed(321)
    again = SimpleNamespace(sdf_func=SDF(opt).to(DEV), color_func=RadF(opt).to(DEV))
    assert util.restore_checkpoint_sfm(opt, again, resume=True) == (None, 41)
    second = check(again.sdf_func, again.color_func)
    for k in first:
        assert torch.equal(first[k], second[k]), k             # same weights, same kernels: same bits


def _batches(n_steps, n_rays, extent):
    out = []
    for it in range(n_steps):
        c, r = _rays(n_rays, extent, 700 + it)
        gt = torch.rand(2, n_rays // 2, 3, generator=torch.Generator().manual_seed(800 + it)).to(DEV)
        out.append((c.view(2, n_rays // 2, 3).contiguous(), r.view(2, n_rays // 2, 3).contiguous(), gt))
    return out


@pytest.mark.parametrize("capture,warm", [(False, False), (True, False), (True, True)])
def test_resumed_stage_continues_the_uninterrupted_trajectory(capture, warm, tmp_path):
    """6 steps in one go == 3 steps, checkpoint (fields + `optim_fields` as the reference stores its optimizers), restore into a
    NEW stage, 3 more steps.  warm: the restoring stage has already captured its hipGraph and stepped on other data -- the loaded
    moments, step count and learning rate must land in the tensors that graph reads (FusedAdam.load_state_dict is in place), and
    the interleaved table copy the graph trusts must be rebuilt from the loaded tables."""
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=32,
                       hash_encoding=dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16))
    opt.output_path = str(tmp_path)
    w = dict(rgb=3, eikonal_loss=1, DC_Loss=0)
    kw = dict(weights=w, lr=4e-3, lr_end=4e-4, max_iter=6, lr_color=2e-3, eps=1e-15)
    n_rays = 128
    data = _batches(6, n_rays, float(opt.data.bound_max[0]))
    # ---- the run that is never interrupted
    sdf_a, rad_a, ren = _randomized(opt, 61)
    st_a = stage.RenderStage(opt, ren, sdf_a, rad_a, capture=capture, **kw)
    ref_losses = [float(st_a.step(*b)["loss_all"]) for b in data]
    # ---- three steps, then a checkpoint
    sdf_b, rad_b, _ = _randomized(opt, 61)
    st_b = stage.RenderStage(opt, ren, sdf_b, rad_b, capture=capture, **kw)
    head = [float(st_b.step(*b)["loss_all"]) for b in data[:3]]
    assert head == ref_losses[:3]
    writer = SimpleNamespace(sdf_func=sdf_b, color_func=rad_b, optim_fields=st_b.optim)
    util.save_checkpoint_sfm(opt, writer, ep=None, it=3, latest=True)
    # ---- a new process's worth of objects: other initial weights, a new optimizer
    sdf_c, rad_c, _ = _randomized(opt, 99)
    st_c = stage.RenderStage(opt, ren, sdf_c, rad_c, capture=capture, **kw)
    if warm:
        junk = _batches(2, n_rays, float(opt.data.bound_max[0]))
        for b in junk:
            st_c.step(*b)                                      # the hipGraph exists and holds the optimizer's addresses
        assert st_c._graph is not None
    reader = SimpleNamespace(sdf_func=sdf_c, color_func=rad_c, optim_fields=st_c.optim)
    assert util.restore_checkpoint_sfm(opt, reader, resume=True) == (None, 3)
    for (k, a), b in zip(sdf_c.state_dict().items(), sdf_b.state_dict().values()):
        assert torch.equal(a, b), k
    assert abs(st_c.optim.param_groups[0]["lr"] - st_b.optim.param_groups[0]["lr"]) < 1e-15
    tail = [float(st_c.step(*b)["loss_all"]) for b in data[3:]]
    torch.cuda.synchronize()
    assert int(st_c.optim.state[st_c.params[0]]["step"]) == 6
    assert abs(st_c.optim.param_groups[1]["lr"] - st_a.optim.param_groups[1]["lr"]) < 1e-12
    # same state, same inputs, same (deterministic) kernels: the resumed run IS the uninterrupted one
    assert tail == ref_losses[3:], (tail, ref_losses[3:])
    for (k, pa), (_, pc) in zip(list(sdf_a.named_parameters()) + list(rad_a.named_parameters()),
                                list(sdf_c.named_parameters()) + list(rad_c.named_parameters())):
        assert torch.equal(pa.detach(), pc.detach()), k
    if st_c.optim._sched:                                      # device schedule == host mirror == the uninterrupted run's
        for gi, t in st_c.optim._sched.items():
            assert torch.equal(t.cpu()[:3], st_a.optim._sched[gi].cpu()[:3]), gi


def test_idle_parameter_group_still_decays_its_rate():
    """ExponentialLR.step() decays EVERY group each iteration; a group whose tensors all have grad None takes no Adam step (torch
    skips such tensors) but its rate moves -- round-3 advisor finding on FusedAdam(scheduled_gamma=...)"""
    from ls2fm.optim import FusedAdam
    a = torch.nn.Parameter(torch.ones(8, device=DEV))
    b = torch.nn.Parameter(torch.ones(8, device=DEV))
    ra, rb = torch.nn.Parameter(torch.ones(8, device=DEV)), torch.nn.Parameter(torch.ones(8, device=DEV))
    ours = FusedAdam([dict(params=[a], lr=1e-2), dict(params=[b], lr=1e-3)], scheduled_gamma=0.5)
    ref = torch.optim.Adam([dict(params=[ra], lr=1e-2), dict(params=[rb], lr=1e-3)])
    sched = torch.optim.lr_scheduler.ExponentialLR(ref, 0.5)
    for it in range(4):
        a.grad = torch.full_like(a, 0.1 * (it + 1)); ra.grad = a.grad.clone()
        if it >= 2:
            b.grad = torch.full_like(b, -0.2); rb.grad = b.grad.clone()
        else:
            b.grad = None; rb.grad = None
        ours.step(); ref.step(); sched.step()
        for g_o, g_r in zip(ours.param_groups, ref.param_groups):
            assert abs(g_o["lr"] - g_r["lr"]) < 1e-15
    assert torch.allclose(a, ra, atol=1e-7) and torch.allclose(b, rb, atol=1e-7)
    assert int(ours.state[b]["step"]) == 2 and float(ours._sched[1].cpu()[1]) == pytest.approx(ref.param_groups[1]["lr"])
