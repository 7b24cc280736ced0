#!/usr/bin/env python
"""Caller-level golden vectors of the POINT side of a bundle-adjustment step (pipelines/BA.py:117-125 and
the "sfm" branch of BA.compute_loss, BA.py:199-202): the reference's own

    xyzs_new, normals_value = sdf_func.get_surface_pts(xyzs)        (models/SDF.py:95-100)
    sdfs = sdf_func.infer_sdf(xyzs_new, mode="ret_sdf").view(-1, 1)
    mask_surf = abs(sdfs) < 2 * sdf_threshold                       (BA.py:116, 131)
    loss = BA.compute_loss(ret)  (mode "sfm": sdf_surf, eikonal on the gradient norms)  ->  BA.summarize_loss  ->  backward

imported from /root/reference where they lie (stubs as in make_golden_caller.py).  The key-point reprojection term is
camera-side logic outside this path: given as a zero with weight None.  Build container only.

    python tests/golden/make_golden_points_step.py   ->  tests/golden/points_step_<case>.npz   (data only)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

CASES = [
    # name, dataset, L, log2_T, n_points
    ("points_step_dtu", "DTU", 8, 12, 300),
    ("points_step_eth3d", "ETH3D", 6, 11, 257),
]
WEIGHTS = dict(reproj_error=None, eikonal_loss=2, sdf_surf=1)


def main():
    from make_golden_caller import import_reference_pipelines
    MG, SDF, RadF, Renderer, RefCamera, RefBA = import_reference_pipelines()
    for ci, (name, dataset, L, log2_T, n_pts) in enumerate(CASES):
        torch.manual_seed(8000 + ci)
        gen = torch.Generator().manual_seed(8100 + ci)
        hash_json = MG.write_hash_json(L, log2_T)
        opt = MG.make_opt(dataset, hash_json, False, 16)
        opt.Res = 128
        opt.loss_weight = MG.AttrDict(ba=dict(WEIGHTS))
        sdf = SDF(opt)
        MG.randomize_module(sdf, gen, table_amp=0.05, w_std=0.03)
        s = (opt.data.bound_max[0] - opt.data.bound_min[0]) / 2
        xyzs = ((torch.rand(n_pts, 3, generator=gen) * 2 - 1) * 0.8 * s).requires_grad_(True)
        xyzs_new, normals_value = sdf.get_surface_pts(xyzs)
        sdfs = sdf.infer_sdf(xyzs_new, mode="ret_sdf").view(-1, 1)
        sdf_threshold = (sdf.bound_max.squeeze()[0] - sdf.bound_min.squeeze()[0]) / 10 / opt.Res
        mask_surf = abs(sdfs) < 2 * sdf_threshold
        ret = MG.AttrDict()
        ret.update(MG.AttrDict(sdfs=sdfs, gradients=normals_value))
        ret.reproj_loss = torch.zeros(())
        me = types.SimpleNamespace(mode="sfm")
        loss = RefBA.BA.compute_loss(me, ret)
        terms = {k: float(v.detach()) for k, v in loss.items()}
        loss = RefBA.BA.summarize_loss(me, opt, loss)
        sdf.zero_grad()
        loss.all.backward()
        out = {}
        out.update(MG.sd_np(sdf, "sdf"))
        out.update({"xyzs": xyzs.detach().numpy(), "xyzs_new": xyzs_new.detach().numpy(),
                    "normals_value": normals_value.detach().numpy(), "sdfs": sdfs.detach().numpy(),
                    "mask_surf": mask_surf.numpy(), "sdf_threshold": np.float32(float(sdf_threshold)),
                    "sdf_surf": np.float32(terms["sdf_surf"]), "eikonal_loss": np.float32(terms["eikonal_loss"]),
                    "loss_all": np.float32(loss.all.item()), "grad/xyzs": xyzs.grad.numpy()})
        out.update(MG.grads_np(sdf, "grad/sdf"))
        meta = dict(dataset=dataset, n_levels=L, log2_hashmap_size=log2_T, dual_field=False, n_samples=16,
                    bgcolor=list(opt.data.bgcolor), iters_max_st=int(opt.SDF.VolSDF.iters_max_st), weights=WEIGHTS, Res=128)
        out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        os.unlink(hash_json)
        print(f"[golden] {name}: sdf_surf={terms['sdf_surf']:.6f} eik={terms['eikonal_loss']:.6f} all={loss.all.item():.5f} "
              f"on-surface {int(mask_surf.sum())}/{n_pts}")


if __name__ == "__main__":
    main()
