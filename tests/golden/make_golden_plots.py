#!/usr/bin/env python
"""Golden lattices of the reference's mesh-export sweeps: `get_grid_uniform` and `get_grid` of /root/reference/utils/plots.py
(:325-370) imported from where they lie (skimage / torchvision / trimesh pre-seeded as empty modules -- those two functions
use numpy and torch only; torch.Tensor.cuda made the identity).  Build container only.
    python tests/golden/make_golden_plots.py  ->  tests/golden/plots_lattices.npz   (data only)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    assert os.path.isdir(REF)
    for name in ("skimage", "skimage.measure", "torchvision", "trimesh", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].measure = sys.modules.get("skimage.measure")
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from utils import plots as ref_plots
    out = {}
    g = ref_plots.get_grid_uniform(7, [-0.6, 0.6])
    out["uniform/points"] = g["grid_points"].numpy()
    for a, v in zip("xyz", g["xyz"]):
        out[f"uniform/{a}"] = np.asarray(v)
    gen = torch.Generator().manual_seed(77)
    for tag, scale in (("fit_x", (0.3, 1.0, 0.8)), ("fit_y", (1.0, 0.25, 0.7)), ("fit_z", (0.9, 1.1, 0.2))):
        pts = torch.randn(500, 3, generator=gen) * torch.tensor(scale)
        g = ref_plots.get_grid(pts, 9)
        out[f"{tag}/input"] = pts.numpy()
        out[f"{tag}/points"] = g["grid_points"].numpy()
        out[f"{tag}/length"] = np.float64(g["shortest_axis_length"])
        out[f"{tag}/index"] = np.int64(g["shortest_axis_index"])
        for a, v in zip("xyz", g["xyz"]):
            out[f"{tag}/{a}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "plots_lattices.npz"), **out)
    print("[golden] plots_lattices:", {k: v.shape for k, v in out.items() if k.endswith("points")})


if __name__ == "__main__":
    main()
