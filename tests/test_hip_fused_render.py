"""GPU parity of the FUSED render forward/backward (ls2fm_render_fwd / ls2fm_render_bwd through the C ABI)
against (a) the golden vectors recorded from the reference, (b) the CPU oracle at the reference's full-size
hash configuration, (c) the general composed form, plus size-independent properties at BASELINE's sizes."""
import numpy as np
import pytest
import torch

import losses
from conftest import GOLDEN_CASES, load_golden, rel_err
from helpers import named_grads, product_for
from ls2fm import fused
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from oracle import fields as OF

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-5
GTOL = 1e-4          # north-star bar: 1e-4 relative, fp32


def _beta_ok(got, ref32, exact, fused_path=True, condition=None):
    """d/d beta is ONE scalar: a sum over every sample of terms of both signs (cancellation ~1e3).  The kernel accumulates
    it in fp64, so it is held to the 1e-4 bar against the exactly summed value of the fp32 computation
    (oracle.fields.beta_gradient_exact_sum: fp32 field evaluations, everything beta enters in fp64); the fp32 reference /
    fp32 oracle value carries summation noise and is only required to be no closer to that value than the kernel is."""
    got, ref32, exact = (float(np.asarray(torch.as_tensor(v).detach().cpu()).reshape(-1)[0]) for v in (got, ref32, exact))
    e_kernel = abs(got - exact) / (abs(exact) + 1e-30)
    e_ref32 = abs(ref32 - exact) / (abs(exact) + 1e-30)
    bar = GTOL if fused_path else max(GTOL, 2.0 * e_ref32)      # composed form: fp32 autograd, as good as the reference's
    if condition is not None:       # sum_i |term_i| / |sum_i term_i|: fp32 inputs (beta, 1 / beta, every summand: 6e-8 each) leave
        bar = max(bar, 2.0 * 6e-8 * condition)                  # an uncertainty of ~eps32 x condition whatever the summation
    assert e_kernel < bar, f"d beta: kernel {got!r} vs exact sum {exact!r}: {e_kernel:.2e} (fp32 reference: {e_ref32:.2e})"
    assert abs(got - ref32) / (abs(ref32) + 1e-30) < GTOL + 2.0 * e_ref32


def _exact_beta_grad(cfg, sdf_state, rad_state, center, ray, tgt, nm):
    osd = {k: v.detach().cpu().float() for k, v in sdf_state.items()}
    ord_ = {k: v.detach().cpu().float() for k, v in rad_state.items()}
    tgt64, nm64 = tgt.detach().cpu().double(), nm.detach().cpu().double()
    return OF.beta_gradient_exact_sum(cfg, center.detach().cpu().float(), ray.detach().cpu().float(), osd, ord_,
                                      lambda ret: losses.render_loss(ret, tgt64, nm64))


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_fused_render_vs_reference_golden(case, manifest):
    g = load_golden(case)
    opt, sdf, rad, ren = product_for(manifest[case], g, DEV)
    center = torch.from_numpy(g["center"]).to(DEV).requires_grad_(True)       # pose gradients on: BA.py:153-154
    ray = torch.from_numpy(g["ray"]).to(DEV).requires_grad_(True)
    took_fused = fused.can_render(ren, opt, center, ray, sdf, rad)
    assert took_fused == (case != "dtu_bgsdf")        # the background-sphere min() is served by the composed form
    ret = ren.forward(opt=opt, center=center, ray=ray, SDF_Field=sdf, Rad_Field=rad)
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert tuple(ret[k].shape) == g[f"ret/{k}"].shape
        assert rel_err(ret[k].cpu(), g[f"ret/{k}"]) < TOL, k
    loss = losses.render_loss(ret, torch.from_numpy(g["rgb_target"]).to(DEV), torch.from_numpy(g["nm_dir"]).to(DEV))
    assert abs(loss.item() - float(g["render_loss"])) < 1e-4 * abs(float(g["render_loss"]))
    loss.backward()
    for name, mod in (("sdf", sdf), ("rad", rad)):
        for k, v in named_grads(mod).items():
            if k == "beta":
                from conftest import golden_cfg
                b64 = _exact_beta_grad(golden_cfg(manifest[case]), sdf.state_dict(), rad.state_dict(), center, ray,
                                          torch.from_numpy(g["rgb_target"]), torch.from_numpy(g["nm_dir"]))
                _beta_ok(v, g[f"render_grad/{name}/{k}"], b64, fused_path=took_fused)
                continue
            assert rel_err(v, g[f"render_grad/{name}/{k}"]) < GTOL, (name, k)
    # gradients w.r.t. the camera rays (the reference's own values)
    assert rel_err(center.grad.cpu(), g["d_center"]) < GTOL
    assert rel_err(ray.grad.cpu(), g["d_ray"]) < GTOL


def test_fused_render_vs_reference_full_size_checksums():
    """the reference's own outputs at the shipped L16/F2/T19 configuration (tests/golden/fullsize_dtu_dual.npz): outputs,
    small and pose gradients in full; the 12 M-entry table gradients by checksums + sparse samples"""
    from conftest import check_table_digest, load_fullsize_golden
    g, sd, rd = load_fullsize_golden()
    n = g["ret/sdfs_volume"].shape[2]
    opt = make_options("DTU", device=DEV, dual_field=True, sample_intvs=n)
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    sdf.load_state_dict(sd, strict=True)
    rad.load_state_dict(rd, strict=True)
    center = torch.from_numpy(g["center"]).to(DEV).requires_grad_(True)
    ray = torch.from_numpy(g["ray"]).to(DEV).requires_grad_(True)
    assert fused.can_render(ren, opt, center, ray, sdf, rad)
    ret = ren.forward(opt=opt, center=center, ray=ray, SDF_Field=sdf, Rad_Field=rad)
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert rel_err(ret[k].cpu(), g[f"ret/{k}"]) < TOL, k
    tgt, nm = torch.from_numpy(g["rgb_target"]), torch.from_numpy(g["nm_dir"])
    loss = losses.render_loss(ret, tgt.to(DEV), nm.to(DEV))
    assert abs(loss.item() - float(g["render_loss"])) < 1e-4 * abs(float(g["render_loss"]))
    loss.backward()
    assert rel_err(center.grad.cpu(), g["d_center"]) < GTOL and rel_err(ray.grad.cpu(), g["d_ray"]) < GTOL
    for pre, mod in (("sdf", sdf), ("rad", rad)):
        for k, v in named_grads(mod).items():
            if k.endswith("embedder_obj.params"):
                check_table_digest(v, g, f"table_grad/{pre}", tol=GTOL)
            elif k == "beta":
                cfg = OF.dataset_config("DTU", dual_field=True, sample_intvs=n)
                _beta_ok(v, g["render_grad/sdf/beta"], _exact_beta_grad(cfg, sd, rd, center, ray, tgt, nm))
            else:
                assert rel_err(v, g[f"render_grad/{pre}/{k}"]) < GTOL, (pre, k)


def test_fused_pose_gradients_equal_composed_at_full_grid():
    """d L / d center, d L / d ray of the fused backward vs the general autograd composition (HIP hash-grid op with its
    double backward + torch layers) on the full L16/T19 grids, dual field; parameter gradients unchanged by asking."""
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=48)
    sdf, rad, ren = _randomized(opt, 21)
    center, ray = _rays(64, 5.0, 22)
    tgt, nm = torch.rand(1, 64, 3, device=DEV), torch.tensor([0.2, -0.4, 0.3], device=DEV)
    out = {}
    for form in ("fused", "composed", "fused_nopose"):
        c = center.clone().requires_grad_(form != "fused_nopose")
        r = ray.clone().requires_grad_(form != "fused_nopose")
        sdf.zero_grad(); rad.zero_grad()
        fn = ren.forward_composed if form == "composed" else ren.forward
        losses.render_loss(fn(opt, c, r, sdf, rad), tgt, nm).backward()
        out[form] = (c.grad, r.grad, {**named_grads(sdf), **{"r." + k: v for k, v in named_grads(rad).items()}})
    assert rel_err(out["fused"][0].cpu(), out["composed"][0].cpu()) < GTOL
    assert rel_err(out["fused"][1].cpu(), out["composed"][1].cpu()) < GTOL
    for k, v in out["fused"][2].items():
        assert rel_err(v, out["fused_nopose"][2][k]) < 1e-6, k           # asking for pose gradients changes nothing else


def _randomized(opt, seed):
    torch.manual_seed(seed)
    sdf, rad, ren = SDF(opt).to(DEV), RadF(opt).to(DEV), Renderer(opt)
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in (sdf, rad):
            for name, p in mod.named_parameters():
                if name.endswith("embedder_obj.params"):
                    p.copy_(((torch.rand(p.shape, generator=gen) * 2 - 1) * 0.1).to(DEV))
                if name.endswith("mlp.0.weight_v") and "Rad_dec" not in name:
                    p[:, 3:] = (torch.randn(p[:, 3:].shape, generator=gen) * 0.05).to(DEV)
                if name.endswith("weight_g"):
                    p.mul_((1 + 0.1 * torch.randn(p.shape, generator=gen)).to(DEV))
    return sdf, rad, ren


def _rays(n, s, seed):
    gen = torch.Generator().manual_seed(seed)
    c = torch.tensor([0.0, 0.0, -2.5 * s]).repeat(n, 1) + 0.02 * s * torch.randn(n, 3, generator=gen)
    d = torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1) + 0.15 * torch.randn(n, 3, generator=gen)
    d[:3] = torch.tensor([0.0, 1.0, -0.3])            # misses
    c[3:6] = 0.2 * s * torch.randn(3, 3, generator=gen)   # origins inside the box
    return c.view(1, n, 3).to(DEV), d.view(1, n, 3).to(DEV)


def _oracle_states(sdf, rad):
    return ({k: v.detach().cpu().clone().requires_grad_(True) for k, v in sdf.state_dict().items()},
            {k: v.detach().cpu().clone().requires_grad_(True) for k, v in rad.state_dict().items()})


@pytest.mark.parametrize("ds,dual,n_samples,n_rays", [("ETH3D", True, 32, 48), ("DTU", False, 40, 32)])
def test_fused_render_vs_oracle_full_size_grid(ds, dual, n_samples, n_rays):
    """the reference's full L16/T19 hash config: outputs and every parameter gradient vs the CPU oracle"""
    opt = make_options(ds, device=DEV, dual_field=dual, sample_intvs=n_samples)
    sdf, rad, ren = _randomized(opt, 3)
    s = opt.data.bound_max[0]
    center, ray = _rays(n_rays, s, 4)
    tgt = torch.rand(1, n_rays, 3, generator=torch.Generator().manual_seed(5))
    nm = torch.tensor([0.2, -0.4, 0.7])
    assert fused.can_render(ren, opt, center, ray, sdf, rad)
    ret = ren.forward(opt, center, ray, sdf, rad)
    losses.render_loss(ret, tgt.to(DEV), nm.to(DEV)).backward()

    cfg = OF.dataset_config(ds, dual_field=dual, sample_intvs=n_samples)
    osd, ord_ = _oracle_states(sdf, rad)
    oret = OF.render(cfg, center.cpu(), ray.cpu(), osd, ord_)
    losses.render_loss(oret, tgt, nm).backward()
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert rel_err(ret[k].cpu(), oret[k]) < TOL, k
    for mod, st in ((sdf, osd), (rad, ord_)):
        for k, v in named_grads(mod).items():
            ref = st[k].grad if st[k].grad is not None else torch.zeros_like(st[k])
            if k == "beta":
                _beta_ok(v, ref, _exact_beta_grad(cfg, sdf.state_dict(), rad.state_dict(), center, ray, tgt, nm))
                continue
            assert rel_err(v, ref) < GTOL, k
            if k.endswith("embedder_obj.params"):       # the whole 12 M-entry table gradient, entry by entry
                from conftest import per_element_check
                worst, n_big = per_element_check(v, ref, k)
                print(f"[{ds}] {k}: per-element worst {worst:.2e} over {n_big} entries above the floor")
    assert osd["embed_fn.embedder_obj.params"].grad.abs().max() > 0


@pytest.mark.parametrize("n_samples", [1, 2, 63, 65, 128, 300])
def test_fused_equals_composed_any_sample_count(n_samples):
    """ragged sample counts (not multiples of the wave size, > 256 -> the 512-thread variant, N = 1, 2)"""
    opt = make_options("BlendedMVS", device=DEV, dual_field=True, sample_intvs=n_samples,
                       hash_encoding=dict(n_levels=6, n_features_per_level=2, log2_hashmap_size=12, base_resolution=16))
    sdf, rad, ren = _randomized(opt, 7)
    center, ray = _rays(20, 2.0, 8)
    tgt, nm = torch.rand(1, 20, 3, device=DEV), torch.tensor([0.5, 0.1, -0.3], device=DEV)
    ret_f = ren.forward(opt, center, ray, sdf, rad)
    losses.render_loss(ret_f, tgt, nm).backward()
    g_f = {**{"s." + k: v for k, v in named_grads(sdf).items()}, **{"r." + k: v for k, v in named_grads(rad).items()}}
    sdf.zero_grad(); rad.zero_grad()
    ret_c = ren.forward_composed(opt, center, ray, sdf, rad)
    losses.render_loss(ret_c, tgt, nm).backward()
    g_c = {**{"s." + k: v for k, v in named_grads(sdf).items()}, **{"r." + k: v for k, v in named_grads(rad).items()}}
    for k in ret_f:
        assert tuple(ret_f[k].shape) == tuple(ret_c[k].shape)
        assert rel_err(ret_f[k].cpu(), ret_c[k].cpu()) < TOL, k
    for k in g_f:
        assert rel_err(g_f[k], g_c[k]) < GTOL, k


@pytest.mark.parametrize("dual", [True, False])
def test_backward_repeats_itself_bit_for_bit_at_baseline_size(dual):
    """1024 rays x 128 samples on the full grids: the coarse levels' slabs are split over several workgroups there.  Default
    scatter mode (ls2fm_set_scatter_mode(1)): their 64-bit fixed-point partials are combined by slab_combine_kernel -- integer
    sums, so two backward passes over the same inputs give IDENTICAL gradients, every table entry being the exactly rounded sum
    of its contributions; mode 0 (float atomics into a zeroed range, rounds 1-3) agrees to summation round-off."""
    from ls2fm import _lib
    lib = _lib.load()
    opt = make_options("ETH3D", device=DEV, dual_field=dual, sample_intvs=128)
    sdf, rad, ren = _randomized(opt, 31)
    center, ray = _rays(1024, 5.0, 32)
    tgt, nm = torch.rand(1, 1024, 3, device=DEV), torch.tensor([0.3, -0.2, 0.6], device=DEV)

    def grads():
        sdf.zero_grad(set_to_none=True); rad.zero_grad(set_to_none=True)
        losses.render_loss(ren.forward(opt, center, ray, sdf, rad), tgt, nm).backward()
        torch.cuda.synchronize()
        return {**{"s." + k: v.clone() for k, v in named_grads(sdf).items()}, **{"r." + k: v.clone() for k, v in named_grads(rad).items()}}

    assert lib.ls2fm_get_scatter_mode() == 1
    a, b = grads(), grads()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    try:
        _lib.check(lib.ls2fm_set_scatter_mode(0), "ls2fm_set_scatter_mode")
        c = grads()
    finally:
        _lib.check(lib.ls2fm_set_scatter_mode(1), "ls2fm_set_scatter_mode")
    for k in a:
        assert rel_err(c[k], a[k]) < 1e-6, k
    tab = a["s.embed_fn.embedder_obj.params"]
    assert int((tab != 0).sum()) > 1000


def test_fused_properties_at_baseline_size():
    """1024 rays x 128 samples, dual field, full-size tables (BASELINE.json config 2 shape): size-independent
    properties -- missed rays render the background exactly, opacity in [0,1], depth within [near, far],
    no-grad forward == grad forward, gradient is linear in the upstream (backward(2 L) == 2 backward(L))."""
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=128)
    sdf, rad, ren = _randomized(opt, 11)
    center, ray = _rays(1024, 5.0, 12)
    ret = ren.forward(opt, center, ray, sdf, rad)
    with torch.no_grad():
        ret_ng = ren.forward(opt, center, ray, sdf, rad)
    for k in ret:
        assert torch.equal(ret[k], ret_ng[k]), k
        assert torch.isfinite(ret[k]).all(), k
    assert torch.equal(ret["rgb"][0, :3].cpu(), torch.zeros(3, 3))        # misses: rgb == bgcolor (black) exactly
    assert (ret["rgb"] >= 0).all() and (ret["rgb"] <= 1).all()
    from ls2fm.ops import ray_aabb_intersect
    _, t, _ = ray_aabb_intersect(center.view(-1, 3), ray.view(-1, 3), ren.center.view(1, 3), ren.half_size.view(1, 3), 1)
    near, far = t[:, 0, 0].view(1, -1, 1), t[:, 0, 1].view(1, -1, 1)
    hit = far[..., 0] > 0
    assert (ret["depth_mlp"][hit] >= near[hit] - 1e-4).all() and (ret["depth_mlp"][hit] <= far[hit] + 1e-4).all()
    loss = (ret["rgb"] - 0.5).abs().mean() * 1e3 + (ret["normals"].norm(dim=-1) - 1).abs().mean() * 1e2
    loss.backward(retain_graph=True)
    g1 = named_grads(sdf)
    sdf.zero_grad()
    (2.0 * loss).backward()
    g2 = named_grads(sdf)
    for k in g1:
        assert rel_err(g2[k], 2.0 * g1[k]) < 1e-4, k           # atomics reorder the fp32 sums, nothing else
    assert g1["embed_fn.embedder_obj.params"].abs().max() > 0


def test_fused_sharding_invariance_large_batch():
    """4096 rays x 128 samples (512 k sample points, full-size dual grids): the gradients of a sum-type loss over the
    whole batch equal the sum of the gradients over four 1024-ray shards -- the property the data-parallel path relies
    on, at a size the composed form / the CPU oracle cannot reach in test time."""
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=128)
    sdf, rad, ren = _randomized(opt, 31)
    center, ray = _rays(4096, 5.0, 32)
    tgt = torch.rand(1, 4096, 3, device=DEV)

    def loss_of(ret, sl):
        return ((ret["rgb"] - tgt[:, sl]).abs().sum() + 0.1 * ((ret["normals"].norm(dim=-1) - 1.0) ** 2).sum()
                + 0.01 * ret["depth_mlp"].sum() + 0.05 * (ret["sdfs_volume"] ** 2).sum())

    sdf.zero_grad(); rad.zero_grad()
    loss_of(ren.forward(opt, center, ray, sdf, rad), slice(None)).backward()
    full = {**named_grads(sdf), **{"r." + k: v for k, v in named_grads(rad).items()}}
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in full.items()}
    for q in range(4):
        sl = slice(1024 * q, 1024 * (q + 1))
        sdf.zero_grad(); rad.zero_grad()
        loss_of(ren.forward(opt, center[:, sl].contiguous(), ray[:, sl].contiguous(), sdf, rad), sl).backward()
        for k, v in {**named_grads(sdf), **{"r." + k: v for k, v in named_grads(rad).items()}}.items():
            acc[k] += v.double()
    for k in full:
        assert torch.isfinite(full[k]).all(), k
        assert rel_err(full[k], acc[k]) < (1e-4 if k == "beta" else 2e-5), k


def test_interleaved_dual_table_is_bit_identical_and_tracks_updates(monkeypatch):
    """LS2FM_DUAL_TABLE=version: the forward gathers from an entry-interleaved copy of the two tables.  Same arithmetic in
    the same order -> bit-identical outputs and gradients; the copy follows optimizer steps (torch's and the fused Adam,
    which writes through raw pointers and bumps the version counters itself) and load_state_dict."""
    from ls2fm.optim import FusedAdam
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=64)
    sdf, rad, ren = _randomized(opt, 41)
    center, ray = _rays(96, 5.0, 42)
    tgt, nm = torch.rand(1, 96, 3, device=DEV), torch.tensor([0.1, 0.3, -0.2], device=DEV)

    def run(mode):
        monkeypatch.setattr(fused, "_DUAL_TABLE", mode)
        sdf.zero_grad(); rad.zero_grad()
        ret = ren.forward(opt, center, ray, sdf, rad)
        losses.render_loss(ret, tgt, nm).backward()
        return ({k: ret[k].detach().clone() for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp")},
                {**named_grads(sdf), **{"r." + k: v for k, v in named_grads(rad).items()}})

    def same(a, b):
        for k in a[0]:
            assert torch.equal(a[0][k], b[0][k]), k
        for k in a[1]:
            assert np.array_equal(np.asarray(a[1][k]), np.asarray(b[1][k])), k

    same(run("off"), run("version"))
    params = [p for m in (sdf, rad) for p in m.parameters()]
    for make in (lambda: torch.optim.Adam(params, lr=1e-2), lambda: FusedAdam(params, lr=1e-2)):
        run("version")
        make().step()                       # moves both tables
        same(run("version"), run("off"))
    state = {k: v.clone() for k, v in rad.state_dict().items()}
    with torch.no_grad():
        rad.embed_fn.embedder_obj.params.mul_(0.5)
    moved = run("version")
    same(moved, run("off"))
    rad.load_state_dict(state)
    back = run("version")
    same(back, run("off"))
    assert not torch.equal(moved[0]["rgb"], back[0]["rgb"])


def test_fused_adam_keeps_the_interleaved_copy_current_without_rebuilds():
    """dual field, default mode: the render reads the entry-interleaved copy of the two tables; FusedAdam (plain and scheduled)
    writes the updated table values into that copy in its own pass (ls2fm_adam_step_mirrored), so a training loop triggers NO
    rebuild (ls2fm_interleave_tables) after the first render -- and the copy equals the interleave of the Parameters bit for
    bit after every step; a foreign writer (torch's Adam) makes the next render rebuild it."""
    from ls2fm.optim import FusedAdam
    assert fused._DUAL_TABLE == "version"
    opt = make_options("ETH3D", device=DEV, dual_field=True, sample_intvs=32,
                       hash_encoding=dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16))
    sdf, rad, ren = _randomized(opt, 71)
    center, ray = _rays(64, 5.0, 72)
    tgt, nm = torch.rand(1, 64, 3, device=DEV), torch.tensor([0.1, 0.3, -0.2], device=DEV)
    t1, t2 = sdf.embed_fn.embedder_obj.params, rad.embed_fn.embedder_obj.params
    lib = fused._lib.load()
    calls = {"n": 0}
    real = lib.ls2fm_interleave_tables

    class Counting:
        def __getattr__(self, k):
            return getattr(lib, k)

        def ls2fm_interleave_tables(self, *a):
            calls["n"] += 1
            return real(*a)

    def interleaved():
        return torch.stack([t1.detach().view(-1, 2), t2.detach().view(-1, 2)], dim=1).reshape(-1)

    def step(optim):
        for p in params:
            p.grad = None
        losses.render_loss(ren.forward(opt, center, ray, sdf, rad), tgt, nm).backward()
        optim.step()

    params = [p for m in (sdf, rad) for p in m.parameters()]
    import unittest.mock as mock
    with mock.patch.object(fused._lib, "load", lambda: Counting()):
        for kw in (dict(), dict(scheduled_gamma=0.9)):
            optim = FusedAdam([dict(params=list(sdf.parameters()), lr=1e-2), dict(params=list(rad.parameters()), lr=3e-3)], **kw)
            step(optim)
            before = calls["n"]
            for _ in range(3):
                step(optim)
                mir = t1._ls2fm_mirror[0]
                assert mir is t2._ls2fm_mirror[0] and mir.fresh(t1, t2)
                assert torch.equal(mir.table, interleaved())
            assert calls["n"] == before, "a FusedAdam loop must not rebuild the interleaved copy"
        torch.optim.Adam(params, lr=1e-2).step()                # a writer that knows nothing of the copy
        assert not t1._ls2fm_mirror[0].fresh(t1, t2)
        before = calls["n"]
        ren.forward(opt, center, ray, sdf, rad)
        assert calls["n"] == before + 1 and torch.equal(t1._ls2fm_mirror[0].table, interleaved())


def test_weight_gradient_tail_inside_the_fill_launch_matches_the_side_stream_form(tmp_path):
    """Round 5: with the MLPs' weight gradients contracted inside shade_bwd, the rest of the chain (decoder columns, level-1 sums,
    reduction rows, finalize tasks) runs as leading workgroups of the scatter_fill launch (csrc/side_jobs.h; hand-offs by flags inside
    one launch) when the batch is large enough -- against the side stream of rounds 2-4 (LS2FM_SIDE_IN_FILL=0) and against the
    separate wgrad_mlp launches (LS2FM_FUSED_WGRAD=0): every MLP / decoder / beta gradient of the benchmark's step within 1e-5 of each
    other (the forms differ in summation order only).  The switches are read once per process: three child processes."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, os, numpy as np, torch
root = sys.argv[1]
for p in (root, os.path.join(root, "level-s2fm_official_amd")):
    sys.path.insert(0, p)
import bench
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
opt = make_options("ETH3D", device="cuda", dual_field=True, sample_intvs=128)
torch.manual_seed(0)
sdf, rad, ren = SDF(opt).cuda(), RadF(opt).cuda(), Renderer(opt)
bench.randomize([sdf, rad], seed=0)
c, r = bench.synthetic_rays(512, float(opt.data.bound_max[0]), "cuda", seed=0)
bench.loss_head(ren.forward(opt, c, r, sdf, rad)).backward()
out = {}
for pre, mod in (("s.", sdf), ("r.", rad)):
    for k, p in mod.named_parameters():
        if not k.endswith("embedder_obj.params"):
            out[pre + k] = p.grad.detach().cpu().numpy()
np.savez(sys.argv[2], **out)
'''
    got = {}
    for tag, env in (("fill", {}), ("side", {"LS2FM_SIDE_IN_FILL": "0"}), ("separate", {"LS2FM_FUSED_WGRAD": "0"})):
        path = str(tmp_path / f"{tag}.npz")
        res = subprocess.run([sys.executable, "-c", code, root, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        got[tag] = dict(np.load(path))
    assert len(got["fill"]) >= 20
    for k, ref in got["separate"].items():
        scale = float(np.abs(ref).max()) + 1e-30
        for tag in ("fill", "side"):
            err = float(np.abs(got[tag][k] - ref).max()) / scale
            assert np.isfinite(got[tag][k]).all() and err < 1e-5, (tag, k, err)


@pytest.mark.gpu
def test_table_gradients_do_not_depend_on_the_accumulate_kernels_unit_schedule(tmp_path):
    """Round 5: which units a workgroup of the persistent slab_accumulate kernel takes (tail ids behind the long point-split units,
    no claims behind a unit of the pool's last ids; LS2FM_ACC_HOLD=0: static pairs + claims to the end) only moves work between
    workgroups -- every entry is an exact fixed-point sum, so both table gradients of the benchmark's step and of a 4096-ray step
    (200 of 256 workgroups start on a long unit there) are the same BITS under either schedule.  Two child processes (the switch is
    read once)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, os, numpy as np, torch
root = sys.argv[1]
for p in (root, os.path.join(root, "level-s2fm_official_amd")):
    sys.path.insert(0, p)
import bench
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
opt = make_options("ETH3D", device="cuda", dual_field=True, sample_intvs=128)
torch.manual_seed(0)
sdf, rad, ren = SDF(opt).cuda(), RadF(opt).cuda(), Renderer(opt)
bench.randomize([sdf, rad], seed=0)
out = {}
for rays in (1024, 4096):
    c, r = bench.synthetic_rays(rays, float(opt.data.bound_max[0]), "cuda", seed=0)
    sdf.zero_grad(); rad.zero_grad()
    bench.loss_head(ren.forward(opt, c, r, sdf, rad)).backward()
    out[f"sdf{rays}"] = sdf.embed_fn.embedder_obj.params.grad.detach().cpu().numpy().copy()
    out[f"rad{rays}"] = rad.embed_fn.embedder_obj.params.grad.detach().cpu().numpy().copy()
np.savez(sys.argv[2], **out)
'''
    got = {}
    for tag, env in (("hold", {}), ("plain", {"LS2FM_ACC_HOLD": "0"})):
        path = str(tmp_path / f"{tag}.npz")
        res = subprocess.run([sys.executable, "-c", code, root, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        got[tag] = dict(np.load(path))
    assert len(got["hold"]) == 4
    for k, ref in got["plain"].items():
        assert np.abs(ref).max() > 0 and np.array_equal(got["hold"][k], ref), k
