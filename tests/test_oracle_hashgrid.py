"""The two independent restatements of the tcnn hash grid (torch: oracle/hashgrid.py, C:
oracle/hashgrid_ref.c) must agree -- indices bit-exactly -- and the level geometry must reproduce the
figures of SURVEY.md Appendix A.2.  This op is 'parity unpinned' by the reference (tcnn is not in the
reference tree); agreement of two separately written restatements is the available cross-check."""
import numpy as np
import pytest
import torch

from oracle import c_hashgrid, hashgrid
from oracle.fields import dataset_config


def tricky_points(n, seed=0):
    """uniform points plus: exact cell boundaries, 0, 1, slightly outside [0,1], far outside, negative"""
    g = np.random.default_rng(seed)
    x = g.random((n, 3), dtype=np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2] = (-0.03, 1.04, 0.5)
    x[3] = (-1.7, 2.9, -0.25)
    x[4] = (0.5, 0.25, 0.125)                       # on cell boundaries of the power-of-two levels
    x[5] = np.float32(1.0) - np.float32(2.0 ** -24)
    x[6] = (1e-8, 1 - 1e-7, 0.3333333)
    x[7] = (15.0 / 16.0, 1.0 / 16.0, 8.0 / 16.0)
    return x


@pytest.mark.parametrize("ds", ["DTU", "ETH3D", "BlendedMVS", "scannet"])
def test_level_table_numpy_vs_c_and_survey_figures(ds):
    t = dataset_config(ds).table()
    scale, res, size, offset, hashed = c_hashgrid.level_table(t.n_levels, t.base_resolution, t.per_level_scale,
                                                              t.log2_hashmap_size)
    assert np.array_equal(scale.view(np.uint32), t.scale.view(np.uint32))      # bit-exact float32 scales
    assert np.array_equal(res, t.resolution) and np.array_equal(size, t.size)
    assert np.array_equal(offset, t.offset) and np.array_equal(hashed, t.hashed)
    if ds == "DTU":
        assert list(t.resolution) == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
        assert list(t.size[:5]) == [4096, 12168, 29792, 79512, 205384] and all(s == 1 << 19 for s in t.size[5:])
        assert not t.hashed[:5].any() and t.hashed[5:].all()
    if ds == "ETH3D":
        assert t.resolution[-1] == 10241 and int(t.hashed.sum()) == 12   # float32 scale 10239.006 -> ceil+1


@pytest.mark.parametrize("ds,L,log2_T", [("DTU", 16, 19), ("ETH3D", 16, 19), ("DTU", 8, 10), ("BlendedMVS", 4, 11)])
def test_indices_bit_exact_torch_vs_c(ds, L, log2_T):
    t = dataset_config(ds, n_levels=L, log2_hashmap_size=log2_T).table()
    x = tricky_points(256)
    idx_c, w_c = c_hashgrid.grid_indices(x, t, want_weights=True)
    idx_t = hashgrid.grid_indices(torch.from_numpy(x), t).numpy()
    assert idx_t.shape == idx_c.shape == (256, L, 8)
    assert np.array_equal(idx_t.astype(np.uint32), idx_c)
    assert idx_c.max() < t.size.max()
    for l in range(L):
        assert idx_c[:, l].max() < t.size[l]


def test_values_and_jacobian_torch_vs_c():
    t = dataset_config("DTU", n_levels=8, log2_hashmap_size=12).table()
    g = torch.Generator().manual_seed(3)
    params = (torch.rand(t.n_params, generator=g) * 2 - 1)
    x = torch.from_numpy(tricky_points(128, seed=5)).requires_grad_(True)
    y = hashgrid.encode(x, params, t)
    out_c, dydx_c = c_hashgrid.grid_encode(x.detach().numpy(), params.numpy(), t, want_dy_dx=True)
    assert np.abs(y.detach().numpy() - out_c).max() < 2e-6
    # autograd Jacobian of the torch oracle vs the analytic one in C
    jac = torch.stack([torch.autograd.grad(y[:, c].sum(), x, retain_graph=True)[0] for c in range(y.shape[1])], dim=1)
    denom = np.abs(dydx_c).max()
    assert np.abs(jac.numpy() - dydx_c).max() / denom < 2e-6


def test_second_derivatives_are_mixed_only():
    """pure second partials of a tri-linear interpolant vanish, mixed ones do not"""
    t = dataset_config("DTU", n_levels=3, log2_hashmap_size=10).table()
    g = torch.Generator().manual_seed(4)
    params = (torch.rand(t.n_params, generator=g, dtype=torch.float64) * 2 - 1)
    x = (torch.rand(5, 3, generator=g, dtype=torch.float64) * 0.9 + 0.05).requires_grad_(True)
    y = hashgrid.encode(x, params, t)[:, 2].sum()
    (gx,) = torch.autograd.grad(y, x, create_graph=True)
    (hxx,) = torch.autograd.grad(gx[:, 0].sum(), x)
    assert hxx[:, 0].abs().max() < 1e-12 and hxx[:, 1:].abs().max() > 1e-6
