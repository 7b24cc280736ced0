"""ls2fm_sdf_volume (the device-side lattice sweep behind extract_mesh, utils/util.py:392-425) against SDF.infer_sdf on
the host-built lattice of the reference's own arithmetic: the coordinates are formed in fp64 and rounded to fp32 in the
kernel exactly as numpy does, so the two volumes are bit-identical."""
import numpy as np
import pytest
import torch

from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.utils import util
from ls2fm import fused

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _field(ds, seed, **kw):
    torch.manual_seed(seed)
    opt = make_options(ds, device=DEV, **kw)
    sdf = SDF(opt).to(DEV)
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        sdf.embed_fn.embedder_obj.params.copy_(((torch.rand(sdf.embed_fn.embedder_obj.params.shape, generator=gen) * 2 - 1)
                                                * 0.1).to(DEV))
        w = sdf.SDF_MLP.mlp[0].weight_v
        w[:, 3:] = (torch.randn(w[:, 3:].shape, generator=gen) * 0.05).to(DEV)
    return opt, sdf


@pytest.mark.parametrize("ds,N,bounds,ref_idx", [("DTU", 24, False, True), ("ETH3D", 17, True, True), ("DTU", 16, True, False),
                                                 ("BlendedMVS", 33, False, False)])
def test_volume_equals_infer_sdf_on_the_reference_lattice(ds, N, bounds, ref_idx):
    opt, sdf = _field(ds, 3)
    bmax = [float(v) for v in opt.data.bound_max] if bounds else None
    bmin = [float(v) for v in opt.data.bound_min] if bounds else None
    vol = util.sdf_volume(sdf, volume_size=2.0, N=N, bound_max=bmax, bound_min=bmin, reference_indexing=ref_idx)
    xyz = torch.from_numpy(util.lattice_points(2.0, N, bmax, bmin, reference_indexing=ref_idx)).to(DEV)
    with torch.no_grad():
        want = sdf.infer_sdf(xyz).view(N, N, N)
    assert vol.shape == (N, N, N) and torch.equal(vol, want)
    # a chunk of the sweep is the same slice of the volume
    step, col_origin, _, _ = util._lattice(2.0, N, bmax, bmin)
    part = fused.sdf_volume(sdf, N, step, col_origin, first=5 * N + 3, count=1000, reference_indexing=ref_idx)
    assert torch.equal(part, vol.view(-1)[5 * N + 3: 5 * N + 1003])


def test_full_resolution_sweep_512():
    """the reference's default N = 512 (134 M points, 512 MB of SDF values) in one call; spot-checked against infer_sdf"""
    opt, sdf = _field("ETH3D", 5)
    N = 512
    bmax, bmin = [float(v) for v in opt.data.bound_max], [float(v) for v in opt.data.bound_min]
    vol = util.sdf_volume(sdf, volume_size=2.0, N=N, bound_max=bmax, bound_min=bmin).view(-1)
    assert vol.numel() == N ** 3 and torch.isfinite(vol).all()
    for first in (0, 77 * N * N + 123, N ** 3 - 4096):
        xyz = torch.from_numpy(util.lattice_points(2.0, N, bmax, bmin, first=first, count=4096)).to(DEV)
        with torch.no_grad():
            want = sdf.infer_sdf(xyz).view(-1)
        assert torch.equal(vol[first:first + 4096], want)


def _oracle_sdf(opt, sdf, ds, pts):
    from oracle import fields as OF
    cfg = OF.dataset_config(ds)
    sd = {k: v.detach().cpu() for k, v in sdf.state_dict().items()}
    with torch.no_grad():
        return OF.infer_sdf(pts, sd, cfg, cfg.table())[:, 0]


@pytest.mark.parametrize("ds,N,bounds", [("DTU", 64, False), ("ETH3D", 48, True)])
def test_volume_vs_cpu_oracle_on_lattice_samples(ds, N, bounds):
    """oracle-direct: the device sweep against the CPU oracle's infer_sdf on 3000 seeded lattice positions (points from the
    host restatement of the reference's lattice arithmetic, utils/util.py:399-409), full L16/T19 grid"""
    opt, sdf = _field(ds, 9)
    bmax = [float(v) for v in opt.data.bound_max] if bounds else None
    bmin = [float(v) for v in opt.data.bound_min] if bounds else None
    vol = util.sdf_volume(sdf, volume_size=2.0, N=N, bound_max=bmax, bound_min=bmin).view(-1).cpu()
    pick = torch.randint(0, N ** 3, (3000,), generator=torch.Generator().manual_seed(1))
    xyz = util.lattice_points(2.0, N, bmax, bmin)
    want = _oracle_sdf(opt, sdf, ds, torch.from_numpy(xyz[pick.numpy()]))
    assert float((vol[pick] - want).abs().max()) < 2e-5 * float(want.abs().max())


def test_high_res_mesh_sweeps_vs_cpu_oracle():
    """ls2fm.utils.plots (utils/plots.py:140-222): the uniform 100^3-style sweep and the rotated, box-fitted sweep of
    get_surface_high_res_mesh on the fused kernel; lattice points vs the host formula, values vs the CPU oracle"""
    from ls2fm.utils import plots
    opt, sdf = _field("DTU", 13)
    axes = plots.uniform_axes(40, [-0.6, 0.6])
    z = plots.sdf_on_lattice(sdf.infer_sdf, axes).cpu()
    assert z.numel() == 40 ** 3
    pts = plots.lattice_on_device(axes, "cpu")
    pick = torch.randint(0, z.numel(), (2000,), generator=torch.Generator().manual_seed(2))
    want = _oracle_sdf(opt, sdf, "DTU", pts[pick])
    assert float((z[pick] - want).abs().max()) < 2e-5 * float(want.abs().max())
    # rotated + fitted lattice
    gen = torch.Generator().manual_seed(3)
    cloud = (torch.randn(400, 3, generator=gen) * torch.tensor([0.3, 0.15, 0.25])).to(DEV)
    mean, vecs = plots._principal_frame(cloud)
    assert abs(float(torch.det(vecs)) - 1.0) < 1e-4
    axes2, _, _ = plots.fitted_axes(((cloud - mean) @ vecs.t()).cpu(), 30)
    z2 = plots.sdf_on_lattice(sdf.infer_sdf, axes2, rotation=vecs, offset=mean).cpu()
    world = (plots.lattice_on_device(axes2, DEV) @ vecs + mean).cpu()          # the very points the sweep evaluated
    pick = torch.randint(0, z2.numel(), (2000,), generator=torch.Generator().manual_seed(4))
    want = _oracle_sdf(opt, sdf, "DTU", world[pick])
    assert float((z2[pick] - want).abs().max()) < 2e-5 * float(want.abs().max())
