"""GPU parity of SDF.sphere_tracing at the exit the training stages actually take: K == iters_max on a random
(non-converging) field (models/SDF.py:149-214).

`t += sdf` amplifies a last-bit difference of the field evaluation every trip, so a free-running comparison of two
implementations of the field diverges on a random field.  The loop is therefore pinned in two complementary ways:

  1. **re-synchronised** (exact): the ORACLE's loop -- its masks, the stale-step quirk, the `+` on the far end, the
     clamp, the global break, all pinned against the reference's own outputs by tests/test_oracle_vs_golden.py -- is run
     with the DEVICE's field evaluation plugged into its no-grad loop.  Every trip then starts from identical state on
     both sides, and the HIP tracing kernel must reproduce the oracle's track, far-end history, near/far and trip count
     BIT FOR BIT at K = iters_max (10 / 20 trips); the differentiable tail (d_pred, sdf_last, finish_mask and the
     gradients of a scalar of them) is then compared with the oracle's on that very track.
  2. **free-running** (measured): the pure oracle trace next to the HIP trace; the first trip at which a ray's track
     leaves the oracle's is recorded, and on the rays that never left, d_pred / sdf_last / finish_mask are compared with
     the oracle and with the reference's golden vectors (`st_*`: K = 10 / 20 = iters_max).
"""
import numpy as np
import pytest
import torch

import losses
from conftest import GOLDEN_CASES, golden_cfg, golden_state, load_golden, rel_err
from helpers import named_grads, product_for
from ls2fm import fused
from oracle import fields as OF

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rays(g, extra, s, seed):
    """the golden case's 48 tracing rays + `extra` seeded ones of the same family (incl. misses / inside origins)"""
    c, d = torch.from_numpy(g["st_center"]), torch.from_numpy(g["st_ray"])
    if extra:
        gen = torch.Generator().manual_seed(seed)
        c2 = torch.tensor([0.0, 0.0, -2.5 * s]).repeat(extra, 1) + 0.05 * s * torch.randn(extra, 3, generator=gen)
        d2 = torch.tensor([0.0, 0.0, 1.0]).repeat(extra, 1) + 0.15 * torch.randn(extra, 3, generator=gen)
        d2[:8] = torch.tensor([0.0, 1.0, -0.2]) + 0.05 * torch.randn(8, 3, generator=gen)        # misses
        c2[8:16] = 0.3 * s * torch.randn(8, 3, generator=gen)                                     # origins inside the box
        c, d = torch.cat([c, c2.float()]), torch.cat([d, d2.float()])
    return c.contiguous(), d.contiguous()


_EXTRA_RAYS = dict(wave=1000, wide=3000, narrow=20100)          # + the 48 recorded rays: which kernel ls2fm_sphere_trace picks


def _device_field(sdf):
    def field(q):
        with torch.no_grad():
            return sdf.infer_sdf(q.to(DEV).contiguous(), mode="ret_sdf")[:, 0].cpu()
    return field


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("kernel", ["wave", "wide", "narrow"])
def test_trace_loop_resynchronised_is_bit_exact_at_iters_max(case, kernel, manifest, monkeypatch):
    g = load_golden(case)
    meta = manifest[case]
    opt, sdf, rad, ren = product_for(meta, g, DEV)
    cfg = golden_cfg(meta)
    s = (cfg.bound_max[0] - cfg.bound_min[0]) / 2
    # the wave-per-end kernel serves calls of up to 2048 rays, the 16-lanes-per-end one those up to 20 000, the narrow
    # (lane per ray end) kernel the rest: same code path as production
    c, d = _rays(g, _EXTRA_RAYS[kernel], s, seed=11)
    osd = golden_state(g, "sdf", requires_grad=True)
    det = {}
    od, os_, _, ofin, otrips = OF.sphere_tracing(cfg, c.view(1, -1, 3), d.view(1, -1, 3), osd, rng=False,
                                                 loop_field=_device_field(sdf), details=det)
    assert otrips == cfg.iters_max_st == int(g["st_trips"])          # the K == iters_max exit
    with torch.no_grad():
        near, far, pts, t_hist, k = fused.sphere_trace(sdf, c.to(DEV), d.to(DEV), history=True)
    assert k == otrips
    assert torch.equal(near.cpu(), det["near"]) and torch.equal(far.cpu(), det["far"])
    assert torch.equal(pts.cpu(), det["track"]), "track differs from the oracle loop fed the same field values"
    assert torch.equal(t_hist.cpu(), det["t_end"])
    if kernel != "wave":
        return
    # differentiable tail on that track: d_pred, sdf_last, finish mask, gradients
    losses.tracing_loss(od, os_).backward()
    d_pred, sdf_last, sampled, finish = sdf.sphere_tracing(c.view(1, -1, 3).to(DEV), d.view(1, -1, 3).to(DEV), sdf)
    assert sdf.last_trips == otrips
    assert rel_err(d_pred.cpu(), od) < 1e-4 and rel_err(sdf_last.cpu(), os_) < 1e-4
    extent = cfg.bound_max[0] - cfg.bound_min[0]
    tie = (os_.detach().abs() - extent / 10 / cfg.res).abs() < 1e-6          # |sdf_last| within round-off of the threshold
    assert np.array_equal(finish.cpu().numpy()[~tie.numpy()], ofin.numpy()[~tie.numpy()])
    assert sampled.shape == (1, min(4096, c.shape[0]) * otrips + c.shape[0], 3)
    losses.tracing_loss(d_pred, sdf_last).backward()
    for name, v in named_grads(sdf).items():
        ref = osd[name].grad if osd[name].grad is not None else torch.zeros_like(osd[name])
        assert rel_err(v, ref) < 1e-4, name


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_trace_free_running_vs_oracle_and_reference_goldens(case, manifest, record_property):
    """Where do the device trace and the pure CPU trace part ways?  First-divergence trip per ray (track point more than
    1e-4 of the scene extent away), and full-output parity -- against the oracle and against the reference's own
    K = iters_max outputs -- on the rays that stayed together."""
    g = load_golden(case)
    meta = manifest[case]
    opt, sdf, rad, ren = product_for(meta, g, DEV)
    cfg = golden_cfg(meta)
    extent = cfg.bound_max[0] - cfg.bound_min[0]
    c, d = _rays(g, 0, extent / 2, seed=0)                      # exactly the reference's 48 recorded rays
    osd = golden_state(g, "sdf")
    det = {}
    od, os_, _, ofin, otrips = OF.sphere_tracing(cfg, c.view(1, -1, 3), d.view(1, -1, 3), osd, rng=False, details=det)
    assert otrips == int(g["st_trips"]) == cfg.iters_max_st
    with torch.no_grad():
        near, far, pts, t_hist, k = fused.sphere_trace(sdf, c.to(DEV), d.to(DEV), history=True)
    assert k == otrips
    dev = (pts.cpu() - det["track"]).norm(dim=-1)                # [R,K]
    dev = torch.where(torch.isfinite(dev), dev, torch.zeros_like(dev))      # misses: -1 - d on both sides (inf-free)
    apart = dev > 1e-4 * extent
    first = torch.where(apart.any(dim=1), apart.float().argmax(dim=1), torch.full((c.shape[0],), k))
    together = first == k
    record_property("first_divergence_trip_histogram", np.bincount(first.numpy(), minlength=k + 1).tolist())
    print(f"[{case}] K={k}: rays together through all trips {int(together.sum())}/{c.shape[0]}; "
          f"first-divergence histogram {np.bincount(first.numpy(), minlength=k + 1).tolist()}")
    assert int(together.sum()) >= c.shape[0] // 4, "the comparison below would be vacuous"
    d_pred, sdf_last, sampled, finish = sdf.sphere_tracing(c.view(1, -1, 3).to(DEV), d.view(1, -1, 3).to(DEV), sdf)
    m = together.numpy()
    got_d, got_s, got_f = d_pred.detach().cpu().numpy()[0], sdf_last.detach().cpu().numpy(), finish.cpu().numpy()[:, 0]
    thr = extent / 10 / cfg.res
    for ref_d, ref_s, ref_f, what in ((od.detach().numpy()[0], os_.detach().numpy(), ofin.numpy()[:, 0], "oracle"),
                                      (g["st_d_pred"][0], g["st_sdf_last"], g["st_finish"][:, 0], "reference golden")):
        assert np.allclose(got_d[m], ref_d[m], rtol=2e-3, atol=2e-3 * extent), what     # K sums of a field with |grad| >> 1
        assert np.allclose(got_s[m], ref_s[m], rtol=0, atol=2e-2 * extent), what
        clear = m & (np.abs(np.abs(ref_s) - thr) > 2e-2 * extent)
        assert np.array_equal(got_f[clear], ref_f[clear]), what
    assert tuple(sampled.shape) == tuple(g["st_sampled_shape"])


@pytest.mark.parametrize("case", GOLDEN_CASES[:2])
@pytest.mark.parametrize("kernel", ["wave", "wide", "narrow"])
def test_track_values_from_the_loop_are_the_field_at_the_track_points(case, kernel, manifest):
    """ls2fm_sphere_trace's track_sdf output -- what the static tracing path sums into the depth instead of evaluating the track
    a second time -- is, bit for bit, the field at the track points: also for start ends that stopped being refreshed but kept
    stepping with a stale value (crossed ends, SDF.py:176-183), and for converged ones (whose step is zeroed, not their value)"""
    g = load_golden(case)
    opt, sdf, rad, ren = product_for(manifest[case], g, DEV)
    cfg = golden_cfg(manifest[case])
    s = (cfg.bound_max[0] - cfg.bound_min[0]) / 2
    c, d = _rays(g, _EXTRA_RAYS[kernel], s, seed=23)
    with torch.no_grad():
        near, far, track, t_end, trips = fused.sphere_trace(sdf, c.to(DEV), d.to(DEV), sync=False)
        vals = track._ls2fm_track_sdf
        ref = sdf.infer_sdf(track.reshape(-1, 3).contiguous(), mode="ret_sdf").view(vals.shape)
    assert vals.shape == (c.shape[0], cfg.iters_max_st + 1)
    assert torch.equal(vals, ref), f"{int((vals != ref).sum())} of {vals.numel()} track values differ"
    moved = (track[:, 1:] != track[:, :-1]).any(dim=-1)
    assert bool(moved.any())


@pytest.mark.parametrize("dataset,k_override", [("DTU", None), ("ETH3D", None), ("ETH3D", 0), ("BlendedMVS", 3)])
def test_traced_depth_node_equals_torch_tail(dataset, k_override):
    """ls2fm.fused.traced_depth (ONE node: track evaluation, masked sum over the first K points, clamp at far, last value, finish
    mask, the two masks of Camera.py:515-516; K read from the device) against the same tail written with torch ops on the
    fused point query -- values, masks and every parameter gradient, incl. K = 0 (single current point, SDF.py:201-202), rays
    whose depth is clamped and an upstream on sdf_last"""
    from ls2fm import fused
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    from helpers import named_grads
    dev = "cuda"
    enc = dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16)
    opt = make_options(dataset, device=dev, hash_encoding=enc)
    torch.manual_seed(3)
    sdf = SDF(opt).to(dev)
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for name, p in sdf.named_parameters():
            if name.endswith("embedder_obj.params"):
                p.copy_(((torch.rand(p.shape, generator=gen) * 2 - 1) * 0.05).to(dev))
            if name.endswith("mlp.0.weight_v"):
                p[:, 3:] = (torch.randn(p[:, 3:].shape, generator=gen) * 0.03).to(dev)
    s = float(opt.data.bound_max[0])
    n = 300
    o = torch.tensor([0.0, 0.0, -2.5 * s]).repeat(n, 1).to(dev)
    d = (torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1) + 0.12 * torch.randn(n, 3, generator=gen)).to(dev)
    gt = torch.rand(n, 3, generator=gen).to(dev)
    gt[::5] = 0.99
    with torch.no_grad():
        near, far, track, _, trips = fused.sphere_trace(sdf, o, d, sync=False)
    if k_override is not None:
        trips = torch.full_like(trips, k_override)
    far = far.clone()
    far[::7] = -1e4                                                  # forces the clamp on these rays (d > far)
    cot_d = torch.randn(n, generator=gen).to(dev)
    cot_l = torch.randn(n, generator=gen).to(dev)

    # (a) the fused node
    sdf.zero_grad()
    d_pred, last, finish, mask_bg, mask_dc = fused.traced_depth(sdf, track, trips, near, far, gt)
    ((d_pred * cot_d).sum() + 0.3 * (last * cot_l).sum()).backward()
    g_a = named_grads(sdf)
    # (b) torch ops over the fused point query (what the static tracing path was before the node existed)
    sdf.zero_grad()
    k_max = track.shape[1]
    sdf_tracks = sdf.infer_sdf(track.detach(), mode="ret_sdf")
    k_eff = trips.clamp(min=1).to(torch.int64)
    live = (torch.arange(k_max, device=dev)[None, :, None] < k_eff).to(sdf_tracks.dtype)
    d_ref = (sdf_tracks * live).sum(dim=-2).view(-1) + near
    d_ref = torch.where(d_ref > far, far, d_ref)
    last_ref = sdf_tracks.gather(1, (k_eff - 1).view(1, 1, 1).expand(n, 1, 1))[:, 0, 0]
    ((d_ref * cot_d).sum() + 0.3 * (last_ref * cot_l).sum()).backward()
    g_b = named_grads(sdf)
    extent = sdf.bound_max.reshape(-1)[0] - sdf.bound_min.reshape(-1)[0]
    fin_ref = last_ref.detach().abs() < extent / 10 / opt.Res
    gray = gt.mean(dim=-1)
    bg_ref = (gray < 0.95) & (gray > 0.05)
    assert rel_err(d_pred.cpu(), d_ref.detach().cpu()) < 2e-6 and rel_err(last.cpu(), last_ref.detach().cpu()) < 1e-6
    assert torch.equal(finish.cpu(), fin_ref.cpu())
    assert torch.equal(mask_bg.bool().cpu(), bg_ref.cpu()) and torch.equal(mask_dc.bool().cpu(), (fin_ref & bg_ref).cpu())
    assert torch.equal((d_pred == far).cpu(), (d_ref.detach() == far).cpu()) and bool((d_pred == far).any())     # some rays are clamped
    for k in g_b:
        if k == "beta":
            assert float(g_a[k].abs().max()) == 0.0
            continue
        assert rel_err(g_a[k], g_b[k]) < 2e-5, k


@pytest.mark.parametrize("case", ["tracing_eth3d_inside_false", "tracing_scannet_inside_false"])
@pytest.mark.parametrize("static", [False, True])
def test_free_running_fused_tracing_on_converging_inside_false_field(case, static):
    """FREE-RUNNING comparison on a fixture where it is meaningful for the `inside = False` presets too (ETH3D, ScanNet): cameras
    inside the surface, a near-distance field (tests/golden/make_golden_tracing.py) -- the iteration contracts, so the fused
    kernel (no re-synchronisation, both the host-synchronising and the static-trips form) must land on the REFERENCE's depths,
    last SDF values, finish mask, trip count and parameter gradients at the tight bars of the `inside = True` test
    (test_sphere_tracing_converging_field_vs_reference: rtol 1e-4).  The random-field `st_*` goldens stay a recorded property
    (how many rays stay together), not the parity claim."""
    import json
    from conftest import load_golden, rel_err
    from helpers import named_grads, options_for
    from ls2fm.models.SDF import SDF
    import losses
    g = load_golden(case)
    meta = json.loads(bytes(g["meta_json"]).decode())
    opt = options_for(meta, DEV)
    sdf = SDF(opt).to(DEV)
    sdf.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sdf/")}, strict=True)
    c = torch.from_numpy(g["center"]).to(DEV).view(1, -1, 3)
    d = torch.from_numpy(g["ray"]).to(DEV).view(1, -1, 3)
    d_pred, sdf_last, _, finish = sdf.sphere_tracing(c, d, sdf, static_trips=static)
    trips = int(sdf.last_trips.item()) if static else int(sdf.last_trips)
    assert trips == int(g["trips"])
    assert np.allclose(d_pred.detach().cpu().numpy(), g["d_pred"], rtol=1e-4, atol=1e-5)
    assert float((sdf_last.detach().cpu() - torch.from_numpy(g["sdf_last"])).abs().max()) < 1e-5 * float(opt.data.bound_max[0])
    thr = 2 * float(opt.data.bound_max[0]) / 10 / opt.Res
    tie = np.abs(np.abs(g["sdf_last"]) - thr) < 2e-5                      # rays sitting on the finish threshold
    assert np.array_equal(finish.cpu().numpy().reshape(-1)[~tie], g["finish"].reshape(-1)[~tie])
    losses.tracing_loss(d_pred, sdf_last).backward()
    # gradients flow through sdf(track points): the track positions agree to ~1e-6 of the scene size, and the hash-grid weights
    # of a position move by that times the finest level's scale -- the table gradient (a sum over ~10 track points per ray of
    # those weights) gets the 1e-3 bar of the random-field tracing test, the dense layers 2e-4
    for k, v in named_grads(sdf).items():
        assert rel_err(v, g[f"grad/sdf/{k}"]) < (1e-3 if k.endswith("embedder_obj.params") else 2e-4), k
