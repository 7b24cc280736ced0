#!/usr/bin/env python
"""Caller-level golden vectors (SURVEY 8c "caller-level pins"): the REFERENCE's own `CameraSet.render`
(pipelines/Camera.py:448-538) -- ray picking with a fixed RNG state, Renderer.forward, SDF.sphere_tracing, mask_bg /
mask_finish, rgb_loss / DC_loss / PSNR -- followed by the reference's `BA.compute_loss` + `summarize_loss`
(pipelines/BA.py:186-218) and `loss.all.backward()`, imported from /root/reference where they lie.  Build container only.

    python tests/golden/make_golden_caller.py   ->  tests/golden/caller_<case>.npz   (data only)

Stubs, as in make_golden.py: the two third-party CUDA ops come from the oracle (tinycudann.Encoding, vren.ray_aabb_intersect);
easydict / ipdb / termcolor / plyfile / skimage / open3d / torch_scatter and, here, cv2 / pycolmap / visdom / wis3d / trimesh /
imageio / pyquaternion / lpips / torchvision / utils.util_vis (visualisation, never executed on this path) are empty shims;
torch.Tensor.cuda is the identity.  A CameraSet is assembled without its COLMAP-side constructor: render() with `pose_input`
only reads `cameras[0].intrinsic` and `cameras[0].mesh_grid`.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

CASES = [
    # name, dataset, L, log2_T, dual, N, H, W, rand_rays
    ("caller_dtu_dual", "DTU", 8, 12, True, 24, 24, 32, 96),
    ("caller_eth3d_single", "ETH3D", 6, 11, False, 16, 20, 28, 80),
]
BA_WEIGHTS = dict(reproj_error=None, eikonal_loss=2, sdf_surf=None, rgb=3, DC_Loss=0, tracing_loss=None)   # LevelS2fM.yaml:114-120


def look_at_poses(n, s, gen):
    """n world-to-camera [3,4] poses on a ring around the scene box, looking at its centre"""
    out = []
    for v in range(n):
        ang = 2 * np.pi * v / n + 0.4
        eye = torch.tensor([2.2 * s * np.sin(ang), 0.25 * s, -2.2 * s * np.cos(ang)], dtype=torch.float32)
        eye = eye + 0.05 * s * torch.randn(3, generator=gen)
        fwd = -eye / eye.norm()
        up = torch.tensor([0.0, 1.0, 0.0])
        right = torch.linalg.cross(up, fwd); right = right / right.norm()
        up2 = torch.linalg.cross(fwd, right)
        R = torch.stack([right, up2, fwd], dim=0)                 # world -> camera rotation
        t = -R @ eye
        out.append(torch.cat([R, t[:, None]], dim=1))
    return torch.stack(out)


def import_reference_pipelines():
    """-> (make_golden module, reference SDF, RadF, Renderer classes, pipelines.Camera, pipelines.BA) with the stubs of the
    module docstring installed; cwd is left at the reference root (its option loader reads relative paths)"""
    import make_golden as MG
    assert os.path.isdir(MG.REF)
    MG.install_stubs()

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return lambda *a, **kw: None

    for name in ("cv2", "pycolmap", "visdom", "wis3d", "trimesh", "imageio", "pyquaternion", "lpips", "utils.util_vis", "torchvision"):
        m = _Any(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.path.insert(0, MG.REF)
    os.chdir(MG.REF)
    import warnings
    warnings.filterwarnings("ignore")
    from models.SDF import SDF
    from models.RadF import RadF
    from models.Renderer import Renderer
    from pipelines import Camera as RefCamera
    from pipelines import BA as RefBA
    return MG, SDF, RadF, Renderer, RefCamera, RefBA


def main():
    MG, SDF, RadF, Renderer, RefCamera, RefBA = import_reference_pipelines()

    for ci, (name, dataset, L, log2_T, dual, N, H, W, rand_rays) in enumerate(CASES):
        torch.manual_seed(7000 + ci)
        gen = torch.Generator().manual_seed(7100 + ci)
        hash_json = MG.write_hash_json(L, log2_T)
        opt = MG.make_opt(dataset, hash_json, dual, N)
        opt.H, opt.W = H, W
        opt.camera = MG.AttrDict(model="perspective")
        opt.Renderer = MG.AttrDict(rand_rays=rand_rays)
        opt.loss_weight = MG.AttrDict(ba=dict(BA_WEIGHTS))
        sdf, rad, ren = SDF(opt), RadF(opt), Renderer(opt)
        # a MILD perturbation of the geometric initialisation: the hash path is live (non-zero table gradients) while the field
        # stays close to a sphere SDF, so that sphere tracing converges -- on a strongly random field `t += sdf` amplifies
        # last-bit differences of the field evaluation every trip and d_points / mask_finish (hence DC_loss) would not be
        # reproducible by ANY second implementation (tests/test_hip_sphere_trace_parity.py measures that)
        MG.randomize_module(sdf, gen, table_amp=0.02, w_std=0.01)
        MG.randomize_module(rad, gen)
        s = (opt.data.bound_max[0] - opt.data.bound_min[0]) / 2
        n_views = 2
        poses = look_at_poses(n_views, s, gen)
        focal = 0.9 * W
        intr = torch.tensor([[focal, 0.0, W / 2], [0.0, focal, H / 2], [0.0, 0.0, 1.0]])
        images = torch.rand(n_views, H * W, 3, generator=gen)
        images[:, ::7] = 0.99                                     # some near-white and near-black pixels: outside mask_bg
        images[:, 3::11] = 0.01
        cam0 = types.SimpleNamespace(intrinsic=intr, mesh_grid=RefCamera.camera.mesh_grid(opt))
        cset = RefCamera.CameraSet.__new__(RefCamera.CameraSet)
        cset.opt, cset.cameras, cset.cam_ids = opt, [cam0], [0]
        out = {}
        out.update(MG.sd_np(sdf, "sdf"))
        out.update(MG.sd_np(rad, "rad"))
        seed = 7200 + ci
        torch.manual_seed(seed)
        rays_idx = torch.randperm(H * W)[: rand_rays // n_views]          # the first RNG draw inside render()
        torch.manual_seed(seed)
        ret = MG.AttrDict()
        RefCamera.CameraSet.render(cset, sdf_func=sdf, color_func=rad, Renderer=ren, ret=ret, mode="train", cam_ids=[0, 1],
                                   dp_req=False, pose_input=poses, rgbs_gt=images, pointset=None)
        centers, rays = RefCamera.camera.get_center_and_ray(opt, poses, intr=intr.unsqueeze(0), rays_idx=rays_idx,
                                                            xy_grid=cam0.mesh_grid)
        # the keypoint-side terms of a BA step (reprojection, surface sdf, multi-view tracing: camera / point-set logic outside
        # this path) are given as zeros with weight None: compute_loss / summarize_loss need the keys as 0-dim tensors
        ret.reproj_loss = torch.zeros(())
        ret.sdfs = torch.zeros(4, 1)
        ret.tracing_loss = torch.zeros(())
        me = types.SimpleNamespace(mode="sfm_refine")
        loss = RefBA.BA.compute_loss(me, ret)
        eik = loss.eikonal_loss
        loss = RefBA.BA.summarize_loss(me, opt, loss)
        sdf.zero_grad(); rad.zero_grad()
        loss.all.backward()
        out.update({"poses": poses.numpy(), "intrinsic": intr.numpy(), "images": images.numpy(), "rays_idx": rays_idx.numpy(),
                    "centers": centers.numpy(), "rays": rays.numpy(), "H": np.int32(H), "W": np.int32(W),
                    "rgbs_gt": images[:, rays_idx, :].numpy(),
                    "ret/rgb": ret.rgb.detach().numpy(), "ret/depth_mlp": ret.depth_mlp.detach().numpy(),
                    "ret/normal_mlp": ret.normal_mlp.detach().numpy(), "ret/sdfs_volume": ret.sdfs_volume.detach().numpy(),
                    "ret/normals": ret.normals.detach().numpy(), "mask_bg": ret.mask_bg.numpy(),
                    "rgb_loss": np.float32(ret.rgb_loss.item()), "DC_loss": np.float32(ret.DC_loss.item()),
                    "PSNR": np.float32(ret.PSNR.item()), "eikonal_loss": np.float32(eik.item()),
                    "loss_all": np.float32(loss.all.item())})
        out.update(MG.grads_np(sdf, "grad/sdf"))
        out.update(MG.grads_np(rad, "grad/rad"))
        meta = dict(dataset=dataset, n_levels=L, log2_hashmap_size=log2_T, dual_field=dual, n_samples=N,
                    bgcolor=list(opt.data.bgcolor), iters_max_st=int(opt.SDF.VolSDF.iters_max_st), weights=BA_WEIGHTS)
        out["meta_json"] = np.frombuffer(__import__("json").dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        os.unlink(hash_json)
        out["trips_note"] = np.int32(0)
        print(f"[golden] {name}: rgb_loss={ret.rgb_loss.item():.6f} DC={ret.DC_loss.item():.6f} PSNR={ret.PSNR.item():.4f} "
              f"eik={eik.item():.5f} all={loss.all.item():.4f} mask_bg={int(ret.mask_bg.sum())}/{ret.mask_bg.numel()}")


if __name__ == "__main__":
    main()
