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
