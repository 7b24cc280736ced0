"""GPU parity: the HIP hash-grid kernels (through the C ABI) vs the CPU oracle -- hash indices bit-exact,
values / first derivatives / double backward within fp32 round-off -- and ray/AABB vs the oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from helpers import oracle_table_of
from ls2fm import hashgrid, ops
from oracle import c_hashgrid, hashgrid as o_hash, ray_aabb as o_aabb
from oracle.fields import dataset_config
from test_oracle_hashgrid import tricky_points

pytestmark = pytest.mark.gpu
DEV = "cuda"


def desc_for(ds, L, log2_T):
    t = dataset_config(ds, n_levels=L, log2_hashmap_size=log2_T).table()
    return hashgrid.build_grid_desc(L, 2, log2_T, 16, t.per_level_scale), t


@pytest.mark.parametrize("ds,L,log2_T", [("DTU", 16, 19), ("ETH3D", 16, 19), ("BlendedMVS", 16, 19), ("scannet", 16, 19),
                                         ("DTU", 8, 10), ("BlendedMVS", 4, 11)])
def test_hash_indices_bit_exact(ds, L, log2_T):
    desc, t = desc_for(ds, L, log2_T)
    x = tricky_points(4096, seed=11)
    got = ops.grid_indices(torch.from_numpy(x).to(DEV), desc).cpu().numpy().astype(np.uint32)
    want = c_hashgrid.grid_indices(x, t)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_empty_and_ragged_inputs():
    desc, t = desc_for("DTU", 4, 10)
    table = torch.rand(t.n_params, device=DEV)
    y = ops.grid_encode(torch.zeros(0, 3, device=DEV), table, desc)
    assert y.shape == (0, 8)
    for n in (1, 63, 65, 257):                      # not multiples of the wave / block size
        x = torch.rand(n, 3, device=DEV)
        y = ops.grid_encode(x, table, desc)
        want = o_hash.encode(x.cpu(), table.cpu(), t)
        assert rel_err(y.cpu(), want) < 2e-6
    with pytest.raises(ValueError):
        ops.grid_encode(torch.zeros(5, 2, device=DEV), table, desc)


@pytest.mark.parametrize("ds,L,log2_T", [("DTU", 16, 19), ("ETH3D", 6, 12)])
def test_forward_backward_double_backward_vs_oracle(ds, L, log2_T):
    desc, t = desc_for(ds, L, log2_T)
    g = torch.Generator().manual_seed(5)
    table_c = (torch.rand(t.n_params, generator=g) * 2 - 1) * 0.5
    x_c = torch.from_numpy(tricky_points(2000, seed=3))
    w1 = torch.randn(2 * L, generator=g)
    w2 = torch.randn(3, generator=g)

    def run(x, table, enc):
        x = x.clone().requires_grad_(True)
        table = table.clone().requires_grad_(True)
        y = enc(x, table)
        s = (y * w1.to(y.device)).sum(-1)                          # a scalar field of the encoding
        (gx,) = torch.autograd.grad(s.sum(), x, create_graph=True)  # its gradient (like SDF.gradient)
        loss = ((gx * w2.to(y.device)).sum(-1) ** 2).mean() + 0.1 * (gx.norm(dim=-1) - 1).abs().mean() + y.pow(2).mean()
        loss.backward()
        return y.detach().cpu(), gx.detach().cpu(), x.grad.cpu(), table.grad.cpu(), loss.item()

    y_o, gx_o, dx_o, dt_o, l_o = run(x_c, table_c, lambda x, tb: o_hash.encode(x, tb, t))
    y_h, gx_h, dx_h, dt_h, l_h = run(x_c.to(DEV), table_c.to(DEV), lambda x, tb: ops.grid_encode(x, tb, desc))
    assert rel_err(y_h, y_o) < 2e-6
    assert rel_err(gx_h, gx_o) < 5e-6
    assert abs(l_h - l_o) < 1e-5 * abs(l_o)
    assert rel_err(dt_h, dt_o) < 2e-5          # table gradient incl. the double-backward scatter (atomics: order differs)
    assert rel_err(dx_h, dx_o) < 2e-5          # incl. mixed second partials
    assert dt_o.abs().max() > 0 and dx_o.abs().max() > 0


def test_jacobian_output_matches_c_oracle():
    import ctypes
    from ls2fm import _lib
    desc, t = desc_for("DTU", 8, 12)
    g = torch.Generator().manual_seed(8)
    table = (torch.rand(t.n_params, generator=g) * 2 - 1)
    x = torch.from_numpy(tricky_points(512, seed=2))
    y = torch.empty(512, 16, device=DEV)
    jac = torch.empty(512, 16, 3, device=DEV)
    xd, td = x.to(DEV), table.to(DEV)
    _lib.check(_lib.load().ls2fm_grid_encode_fwd(ctypes.byref(desc), _lib.ptr(xd), _lib.ptr(td), 512, _lib.ptr(y),
                                                 _lib.ptr(jac), _lib.stream_ptr()), "fwd")
    out_c, jac_c = c_hashgrid.grid_encode(x.numpy(), table.numpy(), t, want_dy_dx=True)
    assert np.abs(y.cpu().numpy() - out_c).max() < 2e-6
    assert np.abs(jac.cpu().numpy() - jac_c).max() / np.abs(jac_c).max() < 2e-6


def test_second_order_through_table_gradient():
    """differentiate through dL/dtable (linear in dy): exercised via autograd.grad of a function of table.grad"""
    desc, t = desc_for("DTU", 3, 10)
    g = torch.Generator().manual_seed(9)
    x_c = torch.rand(64, 3, generator=g)
    tb_c = torch.rand(t.n_params, generator=g)
    w = torch.randn(t.n_params, generator=g)

    def run(x, tb, enc, w):
        x = x.clone().requires_grad_(True)
        tb = tb.clone().requires_grad_(True)
        y = enc(x, tb)
        (gt,) = torch.autograd.grad((y ** 2).sum(), tb, create_graph=True)
        ((gt * w).sum()).backward()
        return x.grad.cpu(), tb.grad.cpu()

    dx_o, dt_o = run(x_c, tb_c, lambda x, tb: o_hash.encode(x, tb, t), w)
    dx_h, dt_h = run(x_c.to(DEV), tb_c.to(DEV), lambda x, tb: ops.grid_encode(x, tb, desc), w.to(DEV))
    assert rel_err(dt_h, dt_o) < 2e-5 and rel_err(dx_h, dx_o) < 2e-5


def test_ray_aabb_vs_oracle():
    g = torch.Generator().manual_seed(1)
    n = 5000
    o = torch.randn(n, 3, generator=g) * 2
    d = torch.randn(n, 3, generator=g)
    d[:50, 0] = 0.0                                  # axis-parallel rays (inf / nan slabs)
    d[50:60] = 0.0
    o[60:200] *= 0.2                                 # origins inside the box -> near clamps to 0
    c = torch.tensor([[0.1, -0.2, 0.05]])
    h = torch.tensor([[1.0, 0.7, 1.3]])
    cnt, t, idx = ops.ray_aabb_intersect(o.to(DEV), d.to(DEV), c.to(DEV), h.to(DEV), 1)
    cnt_o, t_o, idx_o = o_aabb.ray_aabb_intersect(o, d, c, h, 1)
    assert isinstance(ops.ray_aabb_intersect(o.to(DEV), d.to(DEV), c.to(DEV), h.to(DEV), 1), list)
    assert torch.equal(cnt.cpu(), cnt_o) and torch.equal(idx.cpu(), idx_o)
    assert torch.equal(t.cpu(), t_o)                  # same IEEE ops in the same order: bit-exact
    assert not t.requires_grad
    assert (t_o[:, 0, 0] == -1).any() and (t_o[:, 0, 0] == 0).any() and (t_o[:, 0, 0] > 0).any()
    # several boxes, max_hits 2: nearest two, sorted by near t
    c3 = torch.tensor([[0.0, 0, 0], [0.0, 0, 3.0], [0.0, 0, -3.0]])
    h3 = torch.ones(3, 3)
    cnt, t, idx = ops.ray_aabb_intersect(o.to(DEV), d.to(DEV), c3.to(DEV), h3.to(DEV), 2)
    cnt_o, t_o, idx_o = o_aabb.ray_aabb_intersect(o, d, c3, h3, 2)
    assert torch.equal(cnt.cpu(), cnt_o) and torch.equal(idx.cpu(), idx_o) and torch.equal(t.cpu(), t_o)
    assert int(cnt_o.max()) >= 2
