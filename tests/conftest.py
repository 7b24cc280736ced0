import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "level-s2fm_official_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


GOLDEN_CASES = ["dtu_single", "eth3d_dual", "bmvs_dual_white", "dtu_bgsdf", "scannet_single"]


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))


def golden_cfg(meta):
    from oracle.fields import dataset_config
    return dataset_config(meta["dataset"], n_levels=meta["n_levels"], log2_hashmap_size=meta["log2_hashmap_size"],
                          dual_field=meta["dual_field"], sample_intvs=meta["n_samples"], bg_sdf=meta["bg_sdf"],
                          bgcolor=tuple(meta["bgcolor"]), iters_max_st=meta["iters_max_st"])


def golden_state(g, prefix, dtype=torch.float32, requires_grad=False):
    """state dict (reference key names) stored under '<prefix>/' in a golden npz"""
    out = {}
    for k, v in g.items():
        if k.startswith(prefix + "/"):
            out[k[len(prefix) + 1:]] = torch.from_numpy(v).to(dtype).clone().requires_grad_(requires_grad)
    return out


def rel_err(a, b):
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def per_element_check(got, ref, what, rtol=2e-3, floor=1e-3, zero_floor=1e-9):
    """PER-ELEMENT bar for table gradients (the max-norm bar `rel_err` says nothing about small entries, and Adam turns the sign /
    zero-ness of a small gradient into a +-lr step): every entry whose reference magnitude is >= `floor` x the largest one is
    held to `rtol` RELATIVE TO ITS OWN MAGNITUDE, and zero-ness must agree entry by entry: where the reference is exactly zero
    the product is at most `zero_floor` x the scale, and where the reference exceeds that the product is non-zero.
    (`zero_floor`: the product accumulates in 64-bit fixed point with a quantum of 2^-41 of the level's largest contribution,
    so contributions below ~1e-13 of the scale -- samples whose transmittance has underflowed; a fifth of the reference's
    non-zero entries are below 1e-18 -- are dropped; Adam's eps = 1e-8 makes any gradient below ~1e-10 a non-step anyway.)
    An entry of a table gradient is a sum of contributions of both signs, so the per-element bar is the summation noise of
    the fp32 reference, not 1e-4.  -> (worst relative error above the floor, number of entries above the floor)"""
    got = torch.as_tensor(got).detach().cpu().double().reshape(-1)
    ref = torch.as_tensor(ref).detach().cpu().double().reshape(-1)
    scale = float(ref.abs().max())
    big = ref.abs() >= floor * scale
    worst = float(((got - ref).abs()[big] / ref.abs()[big]).max()) if bool(big.any()) else 0.0
    assert worst <= rtol, f"{what}: per-element relative error {worst:.2e} above the floor ({int(big.sum())} entries)"
    tiny = zero_floor * scale
    ghost = (ref == 0) & (got.abs() > tiny)                 # the product invents a gradient
    lost = (ref.abs() > tiny) & (got == 0)                  # the product drops one
    assert not bool(ghost.any()) and not bool(lost.any()), \
        f"{what}: zero-ness differs: {int(ghost.sum())} entries non-zero only here, {int(lost.sum())} zero only here (scale {scale:.2e})"
    return worst, int(big.sum())


def load_fullsize_golden():
    """tests/golden/fullsize_dtu_dual.npz (reference outputs at the shipped L16/F2/T19 configuration) + the two seeded
    tables it was recorded with, regenerated and checked against the stored sha256 -> (golden, sdf_state, rad_state)"""
    import hashlib
    from make_golden_fullsize import fullsize_tables
    g = dict(np.load(os.path.join(GOLDEN, "fullsize_dtu_dual.npz")))
    tabs = fullsize_tables(int(g["n_params"]))
    for name, t in tabs.items():
        digest = np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8)
        assert np.array_equal(digest, g[f"table_sha256/{name}"]), f"seeded {name} table differs from the recorded one"
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sdf/")}
    rd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("rad/")}
    sd["embed_fn.embedder_obj.params"] = tabs["sdf"]
    rd["embed_fn.embedder_obj.params"] = tabs["rad"]
    return g, sd, rd


def check_table_digest(grad, g, prefix, tol=1e-4):
    """table gradient (12 M entries) against the recorded checksums and sparse samples"""
    from make_golden_fullsize import probe_vector
    grad = torch.as_tensor(grad).detach().cpu()
    g64 = grad.double()
    pos = torch.from_numpy(g[prefix + "/sample_pos"])
    ref = torch.from_numpy(g[prefix + "/sample_val"])
    scale = float(ref.abs().max())
    assert float((grad[pos] - ref).abs().max()) < tol * scale, prefix
    # ... and per element on the recorded samples (512 largest + 1024 random positions, most of the latter exactly zero):
    # relative to each sample's own magnitude above the floor, zero <=> zero
    worst, n_big = per_element_check(grad[pos], ref, prefix)
    print(f"[table digest] {prefix}: per-element worst {worst:.2e} over {n_big} samples above the floor")
    abs_sum = float(g[prefix + "/abs_sum"])
    assert abs(float(g64.abs().sum()) - abs_sum) < tol * abs_sum, prefix
    assert abs(float(g64.sum()) - float(g[prefix + "/sum"])) < tol * abs_sum * 1e-2 + 1e-6, prefix   # signed sum: cancels
    probe = float((g64 * probe_vector(grad.numel())).sum())
    # dot with a unit-variance probe: |error| <~ tol * ||grad||_2 ; ||grad||_2 <= sqrt(nnz) * max
    bound = tol * float(g64.norm()) * 4
    assert abs(probe - float(g[prefix + "/probe_dot"])) < bound, prefix
    nnz = int((grad != 0).sum())
    assert abs(nnz - int(g[prefix + "/nnz"])) <= max(8, int(2e-4 * int(g[prefix + "/nnz"]))), (prefix, nnz)
