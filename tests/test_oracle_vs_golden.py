"""Pin the CPU oracle (oracle/fields.py) against golden vectors recorded from the REFERENCE's own
classes (tests/golden/make_golden.py).  Pure CPU; the oracle is what the HIP path is later compared
against, so it has to be trusted first."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, golden_cfg, golden_state, load_golden, rel_err
from oracle import fields as F
import losses

TOL = 2e-6          # the oracle repeats the reference's op sequence; differences are summation-order only
GTOL = 2e-5         # gradients accumulate over many samples


def test_fourier_matches_reference():
    g = load_golden("fourier")
    out = F.fourier_embed(torch.from_numpy(g["d"]))
    assert out.shape[-1] == 27
    assert rel_err(out, g["out"]) < 1e-7


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_infer_sdf_gradient_and_double_backward(case, manifest):
    g = load_golden(case)
    cfg = golden_cfg(manifest[case])
    table = cfg.table()
    sd = golden_state(g, "sdf", requires_grad=True)
    pts = torch.from_numpy(g["pts"])
    y, feat = F.infer_sdf(pts.clone(), sd, cfg, table, "ret_all")
    assert rel_err(y, g["pts_sdf"]) < TOL
    assert rel_err(feat, g["pts_feat"]) < TOL
    p_req = pts.clone()
    nrm = F.sdf_gradient(p_req, sd, cfg, table)
    assert rel_err(nrm, g["pts_normal"]) < TOL
    eik = ((nrm.norm(dim=-1) - 1) ** 2).mean() + 0.2 * (nrm * torch.tensor([0.3, -0.5, 0.8])).sum(-1).mean()
    (eik + 0.1 * y.mean() + 0.05 * (feat[..., 1:] ** 2).mean()).backward()
    assert rel_err(p_req.grad, g["pts_dx"]) < GTOL
    for k, v in sd.items():
        ref = g[f"pts_grad/sdf/{k}"]
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert rel_err(got, ref) < GTOL, k
    # the hash path must be live in these fixtures (SURVEY C-12)
    assert np.abs(g["pts_grad/sdf/embed_fn.embedder_obj.params"]).max() > 0


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_surface_pts_sigma_and_geofeat(case, manifest):
    g = load_golden(case)
    cfg = golden_cfg(manifest[case])
    table = cfg.table()
    sd = golden_state(g, "sdf", requires_grad=True)
    surf, nlen = F.get_surface_pts(torch.from_numpy(g["pts"]).clone(), sd, cfg, table)
    assert rel_err(surf, g["surf_pts"]) < TOL
    assert rel_err(nlen, g["surf_nlen"]) < TOL
    a, b = F.forward_ab(sd, cfg)
    assert rel_err(torch.stack([a, b]).flatten(), g["ab"]) < 1e-7
    sig = F.sdf_to_sigma(torch.from_numpy(g["sigma_in"]), a, b)
    assert rel_err(sig, g["sigma_out"]) < 1e-6
    sig.sum().backward()
    assert rel_err(sd["beta"].grad, g["sigma_dbeta"]) < 1e-5
    if cfg.dual_field:
        rad = golden_state(g, "rad")
        gf = F.geometry_feat(torch.from_numpy(g["pts"]), rad, cfg, table)
        assert rel_err(gf, g["pts_geofeat"]) < TOL


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_composite(case, manifest):
    g = load_golden(case)
    rgb_s = torch.from_numpy(g["comp_rgb_s"]).requires_grad_(True)
    sig_s = torch.from_numpy(g["comp_sig_s"]).requires_grad_(True)
    t_s = torch.from_numpy(g["comp_t_s"])
    rgb, prob = F.composite(torch.from_numpy(g["comp_ray"]), rgb_s, sig_s, t_s)
    assert rel_err(rgb, g["comp_rgb"]) < 1e-6
    assert rel_err(prob, g["comp_prob"]) < 1e-6
    n = t_s.shape[2]
    (rgb.sum() + (prob[..., 0] * torch.arange(n - 1)).sum()).backward()
    assert rel_err(rgb_s.grad, g["comp_d_rgb_s"]) < 1e-6
    assert rel_err(sig_s.grad, g["comp_d_sig_s"]) < 1e-5
    # the missed ray (all t == -1) and the zero-length ray contribute nothing
    assert np.all(g["comp_prob"][0, 0] == 0) and np.all(g["comp_prob"][0, 1] == 0)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_render_forward_and_all_gradients(case, manifest):
    g = load_golden(case)
    cfg = golden_cfg(manifest[case])
    sd = golden_state(g, "sdf", requires_grad=True)
    rad = golden_state(g, "rad", requires_grad=True)
    center = torch.from_numpy(g["center"]).requires_grad_(True)
    ray = torch.from_numpy(g["ray"]).requires_grad_(True)
    ret = F.render(cfg, center, ray, sd, rad)
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert ret[k].shape == g[f"ret/{k}"].shape
        assert rel_err(ret[k], g[f"ret/{k}"]) < TOL, k
    loss = losses.render_loss(ret, torch.from_numpy(g["rgb_target"]), torch.from_numpy(g["nm_dir"]))
    assert abs(loss.item() - float(g["render_loss"])) < 1e-4 * abs(float(g["render_loss"]))
    loss.backward()
    assert rel_err(center.grad, g["d_center"]) < GTOL
    assert rel_err(ray.grad, g["d_ray"]) < GTOL
    for name, state in (("sdf", sd), ("rad", rad)):
        for k, v in state.items():
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            assert rel_err(got, g[f"render_grad/{name}/{k}"]) < GTOL, (name, k)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_sphere_tracing(case, manifest):
    g = load_golden(case)
    cfg = golden_cfg(manifest[case])
    sd = golden_state(g, "sdf", requires_grad=True)
    c = torch.from_numpy(g["st_center"]).view(1, -1, 3)
    d = torch.from_numpy(g["st_ray"]).view(1, -1, 3)
    torch.manual_seed(7)
    d_pred, sdf_last, sampled, finish, trips = F.sphere_tracing(cfg, c, d, sd)
    assert trips == int(g["st_trips"])
    assert tuple(sampled.shape) == tuple(g["st_sampled_shape"])
    assert rel_err(d_pred, g["st_d_pred"]) < 1e-5
    assert rel_err(sdf_last, g["st_sdf_last"]) < 1e-5
    assert np.array_equal(finish.numpy(), g["st_finish"])
    losses.tracing_loss(d_pred, sdf_last).backward()
    for k, v in sd.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert rel_err(got, g[f"st_grad/sdf/{k}"]) < 5e-5, k


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_sphere_tracing_converging_branch(case, manifest):
    """geometric-init weights: the loop leaves through 'no unfinished start ray' (trips < iters_max)"""
    g = load_golden(case)
    cfg = golden_cfg(manifest[case])
    cfg.iters_max_st = int(g["st0_iters_max"])
    sd = golden_state(g, "sdf_init", requires_grad=True)
    c = torch.from_numpy(g["st0_center"]).view(1, -1, 3)
    d = torch.from_numpy(g["st0_ray"]).view(1, -1, 3)
    d_pred, sdf_last, _, finish, trips = F.sphere_tracing(cfg, c, d, sd, rng=False)
    assert trips == int(g["st0_trips"])
    # with inside=False the init sdf is negative outside the sphere and t runs away (inf/nan after many
    # trips) -- the reference does the same; compare the finite entries and the non-finite pattern
    got, ref = d_pred.detach().numpy(), g["st0_d_pred"]
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(got))
    assert np.allclose(got[fin], ref[fin], rtol=1e-4, atol=1e-5)
    assert np.array_equal(finish.numpy(), g["st0_finish"])
    if trips < cfg.iters_max_st:      # left through the 'all start rays finished' branch
        assert trips in (17, 24, 144)


def test_state_dict_manifest_and_hash_geometry(manifest):
    """the oracle's parameter containers use the reference's state_dict keys/shapes and the
    level geometry reproduces the externally known tcnn parameter count."""
    gen = torch.Generator().manual_seed(0)
    for dual in (False, True):
        cfg = F.dataset_config("DTU", dual_field=dual)
        ref = manifest["_state_dict_full_dtu"]["dual" if dual else "single"]
        sd = F.init_sdf_state(cfg, gen)
        assert {k: list(v.shape) for k, v in sd.items()} == ref["sdf"]
        rd = F.init_rad_state(cfg, gen)
        assert {k: list(v.shape) for k, v in rd.items()} == ref["rad"]
    geo = manifest["_hash_geometry"]
    for ds, n_params in (("DTU", 12196240), ("BlendedMVS", 12599920), ("scannet", 13074912), ("ETH3D", 13142880)):
        t = F.dataset_config(ds).table()
        assert t.n_params == geo[ds]["n_params"] == n_params      # SURVEY A.2 figures
        assert abs(t.per_level_scale - geo[ds]["per_level_scale"]) < 1e-7
        assert geo[ds]["out_dim"] == 35


def test_render_full_size_grid_vs_reference_checksums():
    """the shipped L16/F2/T19 configuration (DTU bounds, dual field, 128 samples): outputs, small gradients and pose
    gradients in full, the two 12 M-entry table gradients through checksums and sparse samples"""
    from conftest import check_table_digest, load_fullsize_golden
    g, sd, rd = load_fullsize_golden()
    cfg = F.dataset_config("DTU", dual_field=True, sample_intvs=g["ret/sdfs_volume"].shape[2])
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rd = {k: v.clone().requires_grad_(True) for k, v in rd.items()}
    center = torch.from_numpy(g["center"]).requires_grad_(True)
    ray = torch.from_numpy(g["ray"]).requires_grad_(True)
    ret = F.render(cfg, center, ray, sd, rd)
    for k in ("rgb", "sdfs_volume", "normals", "depth_mlp", "normal_mlp"):
        assert rel_err(ret[k], g[f"ret/{k}"]) < TOL, k
    loss = losses.render_loss(ret, torch.from_numpy(g["rgb_target"]), torch.from_numpy(g["nm_dir"]))
    assert abs(loss.item() - float(g["render_loss"])) < 1e-5 * abs(float(g["render_loss"]))
    loss.backward()
    assert rel_err(center.grad, g["d_center"]) < GTOL and rel_err(ray.grad, g["d_ray"]) < GTOL
    for pre, st in (("sdf", sd), ("rad", rd)):
        for k, v in st.items():
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            if k.endswith("embedder_obj.params"):
                check_table_digest(got, g, f"table_grad/{pre}", tol=GTOL)
            else:
                assert rel_err(got, g[f"render_grad/{pre}/{k}"]) < GTOL, (pre, k)


@pytest.mark.parametrize("case", ["tracing_eth3d_inside_false", "tracing_scannet_inside_false"])
def test_sphere_tracing_converging_field_inside_false(case):
    """the `inside = False` datasets with the cameras inside the surface (tests/golden/make_golden_tracing.py): the root-find
    converges, the oracle lands on the reference's depths, masks, trip count and gradients"""
    import json
    g = load_golden(case)
    meta = json.loads(bytes(g["meta_json"]).decode())
    cfg = golden_cfg(meta)
    sd = golden_state(g, "sdf", requires_grad=True)
    c = torch.from_numpy(g["center"]).view(1, -1, 3)
    d = torch.from_numpy(g["ray"]).view(1, -1, 3)
    d_pred, sdf_last, _, finish, trips = F.sphere_tracing(cfg, c, d, sd, rng=False)
    assert trips == int(g["trips"])
    assert rel_err(d_pred, g["d_pred"]) < 1e-5 and np.array_equal(finish.numpy(), g["finish"])
    assert float((sdf_last.detach() - torch.from_numpy(g["sdf_last"])).abs().max()) < 1e-5
    losses.tracing_loss(d_pred, sdf_last).backward()
    for k, v in sd.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert rel_err(got, g[f"grad/sdf/{k}"]) < 5e-5, k
