#!/usr/bin/env python
"""Stage-LOOP golden vectors: K consecutive iterations of the REFERENCE's own stage loops, driven through the
reference's own `Camera` / `CameraSet` / `Point3DSet` objects, `torch.optim.Adam` and `ExponentialLR`, imported from
/root/reference where they lie.  Build container only.

    python tests/golden/make_golden_stage.py   ->  tests/golden/stage_<case>.npz   (data only)

  refine_*   `Refine.run`  (pipelines/rendering_refine.py:72-97): per iteration  CameraSet.render(pose_input, rgbs_gt, pointset) --
             ray pick, multi-view tracing consistency of one random camera's key points (Camera.py:466-476), Renderer.forward,
             SDF.sphere_tracing, masks, rgb / DC losses, PSNR -- compute_loss / summarize_loss (eikonal over ALL normals,
             sdf_surf on the key-point tracks, 10^w weights), backward, Adam.step, ExponentialLR.step
  init_*     `Initializer.run` (pipelines/Initialization.py:139-226), constructed with `cam_info_reloaded` (the two-view pose
             initialisation of its __init__ is pycolmap's essential-matrix estimation: not part of the loop): per iteration the
             matched key points of each view are traced onto the surface and projected into the other view (Camera.proj_cam_i),
             CameraSet.render with the cameras' own poses, compute_loss (re-projection error, sdf_surf on the tracks, eikonal over
             ALL normals), backward, Adam over the two fields; after the loop the two-view triangulation (mean of the two traced
             points, 3-sigma + finish-mask filter).  `essential_2view` (a pycolmap cross-check whose result no loss reads) and the
             image / pose-evaluation outputs after the loop are no-ops here.
  geoinit_*  `Registration.geo_init_nf` (pipelines/Registration.py:133-296): a NEW camera against two registered ones, SDF field
             only.  Per iteration: the matched key points of every (new, registered) pair are traced from both sides in ONE
             sphere_tracing call; for the matches without a 3-D point yet the two traced points are re-projected into the other
             view (outlier masks from the finish flags and 15 / 30 / 60 px bounds), for the matches that have one the traced
             point is compared with it (tracing_loss); sdf_surf over the tracks' last values and the existing points near the
             surface, eikonal over the existing points, the track points and the random along-ray points of sphere_tracing
             (`sampled_pts`: the `torch.rand_like` draw is recorded); backward, Adam, ExponentialLR.  After the loop: which new
             matches become 3-D points, and where (mean of the two traced points).
  ba_*       `BA.run_ba`   (pipelines/BA.py:110-188), mode "sfm_refine", two cameras: the point side (get_surface_pts,
             infer_sdf, re-projection through the pose parameters, mask_surf), the render side, compute_loss (eikonal over
             mask_bg), the adaptive re-projection weight, backward, Adam over poses + both fields, the point update

Recorded per iteration: the RNG draws that pick inputs (the H*W ray permutation's head, the random camera), every loss term,
PSNR; at the end: every parameter (fields, poses), the points.  Inputs: poses, intrinsics, images, key points, tracks, the
initial weights.  Stubs as in make_golden_caller.py (the two third-party CUDA ops come from the oracle).
"""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

K_ITERS = 20
CASES = [
    # name, loop, dataset, L, log2_T, dual, N, H, W, rand_rays, n_kp
    ("stage_refine_dtu_dual", "refine", "DTU", 8, 12, True, 24, 24, 32, 96, 40),
    ("stage_refine_eth3d_single", "refine", "ETH3D", 6, 11, False, 16, 20, 28, 80, 32),
    ("stage_ba_dtu_dual", "ba", "DTU", 8, 12, True, 24, 24, 32, 96, 64),
    ("stage_init_dtu_dual", "init", "DTU", 8, 12, True, 24, 24, 32, 96, 48),
    ("stage_geoinit_dtu", "geoinit", "DTU", 8, 12, True, 24, 24, 32, 96, 40),
]
OPTIM = dict(algo="Adam", algo_split="SGD", optim_split=True, use_grad_clip=False,
             sched=dict(type="ExponentialLR"),
             geoinit=dict(max_iter=K_ITERS // 5, lr_sdf=1e-3, lr_sdf_end=1e-3),           # the loop runs max_iter * 5 iterations
             init=dict(max_iter=K_ITERS, lr_sdf=1e-3, lr_sdf_end=1e-4, lr_color=1e-2, lr_color_end=1e-3),
             refine=dict(max_iter=K_ITERS, lr_sdf=1e-3, lr_sdf_end=5e-4, lr_color=1e-3, lr_color_end=5e-4),
             ba=dict(max_iter=K_ITERS, lr_sdf=1e-4, lr_sdf_end=5e-5, lr_pose=1e-2, lr_pose_end=5e-3, lr_color=1e-3,
                     lr_color_end=5e-4, lr_pose_r=5e-3, lr_pose_t=1e-2))                      # LevelS2fM.yaml:60-90
WEIGHTS = dict(geoinit=dict(reproj_error=0, tracing_loss=1, sdf_surf=2, eikonal_loss=2),
               init=dict(reproj_error=0, eikonal_loss=2, sdf_surf=2, rgb=3, DC_Loss=0),
               refine=dict(eikonal_loss=2, rgb=3, DC_Loss=0, tracing_loss=2, sdf_surf=2),
               ba=dict(reproj_error=0, eikonal_loss=2, sdf_surf=2, rgb=3, DC_Loss=0, tracing_loss=1))   # LevelS2fM.yaml:98-123


def look_out_poses(n, s, gen):
    """n world-to-camera [3,4] poses near the centre of the scene box, looking outwards (the `inside = False` datasets: the
    cameras stand inside the surface, models/SDF.py:66-71)"""
    out = []
    for v in range(n):
        ang = 0.4 + 0.08 * v                                    # a small baseline: the views overlap
        fwd = torch.tensor([np.sin(ang), 0.1, np.cos(ang)], dtype=torch.float32)
        fwd = fwd / fwd.norm()
        eye = 0.05 * s * torch.tensor([np.cos(ang), 0.0, -np.sin(ang)], dtype=torch.float32) * (2 * v - 1) + 0.03 * s * torch.randn(3, generator=gen)
        up = torch.tensor([0.0, 1.0, 0.0])
        right = torch.linalg.cross(up, fwd); right = right / right.norm()
        up2 = torch.linalg.cross(fwd, right)
        R = torch.stack([right, up2, fwd], dim=0)
        out.append(torch.cat([R, (-R @ eye)[:, None]], dim=1))
    return torch.stack(out)


class Recorder:
    """logs the RNG draws that pick a step's inputs while the reference loop runs"""

    def __init__(self, hw):
        self.hw, self.perms, self.cams = hw, [], []
        self._randperm, self._randint = torch.randperm, random.randint

    def __enter__(self):
        def randperm(n, *a, **kw):
            out = self._randperm(n, *a, **kw)
            if n == self.hw:
                self.perms.append(out.clone())
            return out

        def randint(a, b):
            v = self._randint(a, b)
            self.cams.append(v)
            return v
        torch.randperm, random.randint = randperm, randint
        return self

    def __exit__(self, *exc):
        torch.randperm, random.randint = self._randperm, self._randint


def near_poses(n, s, gen):
    """n world-to-camera [3,4] poses on a short arc around the scene box, looking at its centre (neighbouring views: the key
    points of one are visible in the others)"""
    out = []
    for v in range(n):
        ang = 0.4 + 0.22 * v
        eye = torch.tensor([2.2 * s * np.sin(ang), 0.25 * s + 0.1 * s * v, -2.2 * s * np.cos(ang)], dtype=torch.float32)
        eye = eye + 0.03 * s * torch.randn(3, generator=gen)
        fwd = -eye / eye.norm()
        up = torch.tensor([0.0, 1.0, 0.0])
        right = torch.linalg.cross(up, fwd); right = right / right.norm()
        up2 = torch.linalg.cross(fwd, right)
        R = torch.stack([right, up2, fwd], dim=0)
        out.append(torch.cat([R, (-R @ eye)[:, None]], dim=1))
    return torch.stack(out)


def geoinit_case(name, ci, dataset, L, log2_T, dual, N, H, W, n_kp, MG, SDF, RefCamera, RefPoint3D, ref_camera):
    """the scene and the recorded run of the geo_init_nf case (three views; view 2 is the new camera)"""
    from pipelines import Registration as RefReg
    torch.manual_seed(9000 + ci)
    random.seed(9050 + ci)
    gen = torch.Generator().manual_seed(9100 + ci)
    hash_json = MG.write_hash_json(L, log2_T)
    opt = MG.make_opt(dataset, hash_json, dual, N)
    opt.H, opt.W = H, W
    opt.data.image_size = [H, W]
    opt.camera = MG.AttrDict(model="perspective")
    opt.Renderer = MG.AttrDict(rand_rays=64)
    opt.optim = MG.AttrDict(json.loads(json.dumps(OPTIM)))
    opt.loss_weight = MG.AttrDict(json.loads(json.dumps(WEIGHTS)))
    sdf = SDF(opt)
    MG.randomize_module(sdf, gen, table_amp=0.02, w_std=0.01)
    s = (opt.data.bound_max[0] - opt.data.bound_min[0]) / 2
    poses = near_poses(3, s, gen)
    se3 = ref_camera.lie.SE3_to_se3(poses)
    focal = 0.9 * W
    intr = torch.tensor([[focal, 0.0, W / 2], [0.0, focal, H / 2], [0.0, 0.0, 1.0]])
    images = torch.rand(3, 3, H, W, generator=gen)
    # key point j of every view observes (about) the same surface point: traced from the new view, projected into the others
    kp2 = torch.stack([torch.rand(n_kp, generator=gen) * (W - 6) + 3, torch.rand(n_kp, generator=gen) * (H - 6) + 3], dim=-1)
    probe = RefCamera.CameraSet(opt)
    probe.add_camera(id=2, img_gt=images[2], kypts2D=kp2.clone(), pose_gt=poses[2:3], Match_mask=None, Inlier_mask=None, Intrinsic=intr,
                     Extrinsic=se3[2:3])
    with torch.no_grad():
        pts, _, _, _ = probe(2).get_pts3D(sdf, np.arange(n_kp))
        pts = pts[0]
        kps, ok = [], torch.isfinite(pts).all(-1)
        for v in range(2):
            uv = ref_camera.cam2img(ref_camera.world2cam(pts[None], poses[v:v + 1]), intr[None])[0]
            uv = uv[:, :2] / uv[:, 2:] + 0.4 * torch.randn(n_kp, 2, generator=gen)
            ok &= (uv[:, 0] > 1) & (uv[:, 0] < W - 1) & (uv[:, 1] > 1) & (uv[:, 1] < H - 1)
            kps.append(uv)
    assert int(ok.sum()) >= n_kp // 2, f"too few three-view tracks: {int(ok.sum())}/{n_kp}"
    kp = [kps[0][ok].clone(), kps[1][ok].clone(), kp2[ok].clone()]
    pts = pts[ok]
    n = kp[2].shape[0]
    n_exist = n // 2                                                    # the first half already has 3-D points (seen by views 0, 1)
    xyz = pts[:n_exist] + 0.01 * s * torch.randn(n_exist, 3, generator=gen)
    ident = np.stack([np.arange(n), np.arange(n)], axis=-1).astype(np.int32)
    inl = np.ones(n, bool); inl[2::9] = False                           # a few matches flagged as outliers
    cset = RefCamera.CameraSet(opt)
    pset = RefPoint3D.Point3DSet(opt)
    idx3d = -np.ones(n, int); idx3d[:n_exist] = np.arange(n_exist)
    for v in range(3):
        cset.add_camera(id=v, img_gt=images[v], kypts2D=kp[v].clone(), pose_gt=poses[v:v + 1],
                        Match_mask=[ident.copy(), ident.copy()], Inlier_mask=[inl.copy(), inl.copy()], Intrinsic=intr,
                        Extrinsic=se3[v:v + 1], idx2d_to_3d=idx3d.copy())
    for j in range(n_exist):
        pset.add_point3d(xyz[j:j + 1].clone(), [(0, j), (1, j)])
    out = {}
    out.update(MG.sd_np(sdf, "sdf0"))
    out.update({"poses": poses.numpy(), "se3": se3.numpy(), "intrinsic": intr.numpy(), "images": images.numpy(),
                "kypts": torch.stack(kp).numpy(), "xyzs": xyz.numpy(), "H": np.int32(H), "W": np.int32(W), "inliers": inl,
                "n_exist": np.int32(n_exist)})
    reg = RefReg.Registration(opt, sdf, cset)
    reg.src_cam_id = [0, 1]
    log = {k: [] for k in ("all", "reproj_error", "tracing_loss", "sdf_surf", "eikonal_loss")}
    summarize_orig = reg.summarize_loss

    def summarize_logged(o, loss):
        loss = summarize_orig(o, loss)
        for k in log:
            log[k].append(float(loss[k]) if k in loss else float("nan"))
        return loss
    reg.summarize_loss = summarize_logged
    draws = []
    rand_like_orig = torch.rand_like

    def rand_like(t, *a, **kw):
        u = rand_like_orig(t, *a, **kw)
        draws.append(u.clone())
        return u
    torch.rand_like = rand_like
    try:
        reg.geo_init_nf(cset(2), sdf, pset)
    finally:
        torch.rand_like = rand_like_orig
    assert len(log["all"]) == K_ITERS, len(log["all"])
    out["sample_u"] = torch.stack([u.reshape(-1) for u in draws[-K_ITERS:]]).numpy()        # the loop's own tracing calls are the last ones
    for k, v in log.items():
        out[f"log/{k}"] = np.asarray(v, np.float64)
    out.update(MG.sd_np(sdf, "sdf_final"))
    # the block after the loop (Registration.py:271-294): every (new, registered) pair adds ITS OWN points for the matches it keeps
    # (the same key point of the new view can get a point from each pair; its index ends up at the last pair's) -- recorded per
    # new point: the pair's registered view, the new view's key point, the coordinates
    n_pts = len(pset.pointset)
    tracks = pset.get_feat_tracks(idxs=list(range(n_exist, n_pts)))
    out["tri_src_view"] = np.asarray([int(t[1][0]) for t in tracks], np.int64)
    out["tri_kp_new"] = np.asarray([int(t[0][1]) for t in tracks], np.int64)
    out["tri_xyzs"] = (torch.cat(pset.get_xyzs(idxs=list(range(n_exist, n_pts))), dim=0).numpy() if n_pts > n_exist
                       else np.zeros((0, 3), np.float32))
    new_ids = list(range(n_exist, n_pts))
    meta = dict(dataset=dataset, n_levels=L, log2_hashmap_size=log2_T, dual_field=dual, n_samples=N, loop="geoinit", iters=K_ITERS,
                optim=OPTIM["geoinit"], weights=WEIGHTS["geoinit"], bgcolor=list(opt.data.bgcolor),
                iters_max_st=int(opt.SDF.VolSDF.iters_max_st), Res=int(opt.Res), rand_rays=64)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    os.unlink(hash_json)
    print(f"[golden] {name}: matches={n} (existing {n_exist}) loss {log['all'][0]:.4f} -> {log['all'][-1]:.4f}  reproj {log['reproj_error'][0]:.3f} -> "
          f"{log['reproj_error'][-1]:.3f}  new points {len(new_ids)}")


def main():
    import make_golden_caller as MC
    MG, SDF, RadF, Renderer, RefCamera, RefBA = MC.import_reference_pipelines()
    from pipelines import rendering_refine as RefRefine
    from pipelines import Point3D as RefPoint3D
    import utils.camera as ref_camera

    only = set(sys.argv[1:])                       # optional: case names to (re)generate
    for ci, (name, loop, dataset, L, log2_T, dual, N, H, W, rand_rays, n_kp) in enumerate(CASES):
        if only and name not in only:
            continue
        if loop == "geoinit":
            geoinit_case(name, ci, dataset, L, log2_T, dual, N, H, W, n_kp, MG, SDF, RefCamera, RefPoint3D, ref_camera)
            continue
        torch.manual_seed(9000 + ci)
        random.seed(9050 + ci)
        gen = torch.Generator().manual_seed(9100 + ci)
        hash_json = MG.write_hash_json(L, log2_T)
        opt = MG.make_opt(dataset, hash_json, dual, N)
        opt.H, opt.W = H, W
        opt.data.image_size = [H, W]
        opt.camera = MG.AttrDict(model="perspective")
        opt.Renderer = MG.AttrDict(rand_rays=rand_rays)
        opt.optim = MG.AttrDict(json.loads(json.dumps(OPTIM)))
        opt.loss_weight = MG.AttrDict(json.loads(json.dumps(WEIGHTS)))
        sdf, rad, ren = SDF(opt), RadF(opt), Renderer(opt)
        MG.randomize_module(sdf, gen, table_amp=0.02, w_std=0.01)      # mild: sphere tracing converges (make_golden_caller.py)
        MG.randomize_module(rad, gen)
        s = (opt.data.bound_max[0] - opt.data.bound_min[0]) / 2
        n_views = 2
        poses = MC.look_at_poses(n_views, s, gen) if opt.data.inside else look_out_poses(n_views, s, gen)    # world -> camera [V,3,4]
        se3 = ref_camera.lie.SE3_to_se3(poses)                         # [V,6]: what a Camera keeps (Camera.py:76-83)
        focal = 0.9 * W
        intr = torch.tensor([[focal, 0.0, W / 2], [0.0, focal, H / 2], [0.0, 0.0, 1.0]])
        images = torch.rand(n_views, 3, H, W, generator=gen)           # Camera.render reads img_gt.view(3, -1)
        flat = images.view(n_views, 3, -1)
        flat[:, :, ::7] = 0.99                                         # some near-white / near-black pixels: outside mask_bg
        flat[:, :, 3::11] = 0.01

        # ---- cameras and tracked points: key points of camera 0 are traced onto the surface, the points are perturbed and
        # projected into camera 1 (+ noise) -> two-view feature tracks with a small re-projection error
        cset = RefCamera.CameraSet(opt)
        kp0 = torch.stack([torch.rand(n_kp, generator=gen) * (W - 4) + 2, torch.rand(n_kp, generator=gen) * (H - 4) + 2], dim=-1)
        for v in range(n_views):
            cset.add_camera(id=v, img_gt=images[v], kypts2D=kp0.clone(), pose_gt=poses[v:v + 1], Match_mask=None, Inlier_mask=None,
                            Intrinsic=intr, Extrinsic=se3[v:v + 1])
        with torch.no_grad():
            pts0, fin0, _, _ = cset(0).get_pts3D(sdf, np.arange(n_kp))                    # [1,n,3]
            xyz = pts0[0] + 0.01 * s * torch.randn(n_kp, 3, generator=gen)
            uv1 = ref_camera.cam2img(ref_camera.world2cam(xyz[None], poses[1:2]), intr[None])[0]
            uv1 = uv1[:, :2] / uv1[:, 2:] + 0.3 * torch.randn(n_kp, 2, generator=gen)
            ok = (uv1[:, 0] > 1) & (uv1[:, 0] < W - 1) & (uv1[:, 1] > 1) & (uv1[:, 1] < H - 1) & torch.isfinite(xyz).all(-1)
        assert int(ok.sum()) >= n_kp // 4, f"too few two-view tracks: {int(ok.sum())}/{n_kp}"
        kp = [kp0[ok].clone(), uv1[ok].clone()]
        xyz = xyz[ok]
        n_pts = xyz.shape[0]
        pset = RefPoint3D.Point3DSet(opt)
        for v in range(n_views):
            cset(v).kypts = kp[v]
            cset(v).idx2d_to_3d = np.arange(n_pts)
        for j in range(n_pts):
            pset.add_point3d(xyz[j:j + 1].clone(), [(0, j), (1, j)])

        out = {}
        out.update(MG.sd_np(sdf, "sdf0"))
        out.update(MG.sd_np(rad, "rad0"))
        out.update({"poses": poses.numpy(), "se3": se3.numpy(), "intrinsic": intr.numpy(), "images": images.numpy(),
                    "kypts": torch.stack(kp).numpy(), "xyzs": xyz.numpy(), "H": np.int32(H), "W": np.int32(W)})

        # ---- the loop, logged through instance-level wrappers (the loop code itself is the reference's)
        log = {k: [] for k in ("all", "PSNR", "rgb_loss", "DC_loss", "eikonal_loss", "sdf_surf", "tracing_loss", "reproj_error",
                               "w_reproj", "mask_bg_count")}
        render_orig = cset.render

        def render_logged(*a, **kw):
            ret = render_orig(*a, **kw)
            log["PSNR"].append(float(ret.PSNR)); log["rgb_loss"].append(float(ret.rgb_loss)); log["DC_loss"].append(float(ret.DC_loss))
            log["mask_bg_count"].append(int(ret.mask_bg.sum()))
            return ret
        cset.render = render_logged

        if loop == "refine":
            stage = RefRefine.Refine(opt, cset, pset, sdf, rad)                 # consumes one ray permutation (rgbs_gt)
            wkey = "refine"
        elif loop == "init":
            # a fresh camera set / point set, filled by the Initializer itself from `var` (matches + inlier flags per view pair)
            import tempfile
            from pipelines import Initialization as RefInit
            opt.Ablate_config.sdf_filter = True
            opt.output_path = tempfile.mkdtemp(prefix="ls2fm_init_")
            n_all = kp[0].shape[0]
            n_out = max(2, n_all // 8)                                          # a few matches flagged as outliers
            m01 = np.stack([np.arange(n_all), np.arange(n_all)], axis=-1).astype(np.int32)
            inl = np.ones(n_all, bool); inl[np.arange(n_out) * 3 + 1] = False
            var = MG.AttrDict(indx_init=[0, 1], imgs_init=images, poses_gt=poses, kypts_init=kp, intrs_init=[intr, intr],
                              mchs_init=[[m01.copy()], [m01[:, ::-1].copy()]], inliers_init=[[inl.copy()], [inl.copy()]])
            cset = RefCamera.CameraSet(opt)
            pset = RefPoint3D.Point3DSet(opt)
            stage = RefInit.Initializer(opt, cset, pset, sdf, rad, var,
                                        cam_info_reloaded=dict(pose_para=se3, idx2d_to_3ds=[None, None]))
            stage.essential_2view = lambda *a, **kw: None
            cset.eval_poses = lambda *a, **kw: None
            render_orig = cset.render
            cset.render = render_logged
            out["matches"], out["inliers"] = m01, inl
            wkey = "init"
        else:
            stage = RefBA.BA(opt, cset, pset, sdf, rad, cam_pick_ids=None, mode="sfm_refine")
            wkey = "ba"
        summarize_orig = stage.summarize_loss

        def summarize_logged(o, loss):
            if wkey == "ba":
                log["w_reproj"].append(float(o.loss_weight.ba.reproj_error))
            loss = summarize_orig(o, loss)
            for k, dst in (("eikonal_loss", "eikonal_loss"), ("sdf_surf", "sdf_surf"), ("tracing_loss", "tracing_loss"),
                           ("reproj_error", "reproj_error"), ("all", "all")):
                if k in loss:
                    log[dst].append(float(loss[k]))
            return loss
        stage.summarize_loss = summarize_logged
        with Recorder(H * W) as rec:
            if loop == "refine":
                stage.run(sdf, rad, ren)
            elif loop == "init":
                stage.run(cset, pset, sdf, rad, ren)
            else:
                stage.run_ba(sdf, rad, ren)
        assert len(rec.perms) == K_ITERS and len(log["all"]) == K_ITERS, (len(rec.perms), len(log["all"]))
        n_pick = rand_rays // n_views
        out["rays_idx"] = torch.stack([p[:n_pick] for p in rec.perms]).numpy()
        out["cam_pick"] = np.asarray(rec.cams, np.int32)
        for k, v in log.items():
            if v:
                out[f"log/{k}"] = np.asarray(v, np.float64)
        out.update(MG.sd_np(sdf, "sdf_final"))
        out.update(MG.sd_np(rad, "rad_final"))
        if loop == "init":
            idx0 = cset(0).idx2d_to_3d
            out["tri_kept"] = (idx0 != -1)                                      # per key point of view 0: triangulated
            out["tri_xyzs"] = torch.cat(pset.get_xyzs(idxs=list(idx0[idx0 != -1])), dim=0).numpy()
        if loop == "ba":
            out["se3_final"] = torch.cat([stage.r_paras(), stage.t_paras()], dim=1).detach().numpy()
            out["xyzs_final"] = stage.xyzs_all.detach().numpy()
        meta = dict(dataset=dataset, n_levels=L, log2_hashmap_size=log2_T, dual_field=dual, n_samples=N, loop=loop, iters=K_ITERS,
                    rand_rays=rand_rays, optim=OPTIM[wkey], weights=WEIGHTS[wkey], bgcolor=list(opt.data.bgcolor),
                    iters_max_st=int(opt.SDF.VolSDF.iters_max_st), Res=int(opt.Res))
        out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        os.unlink(hash_json)
        print(f"[golden] {name}: points={n_pts} loss {log['all'][0]:.4f} -> {log['all'][-1]:.4f}  PSNR {log['PSNR'][0]:.3f} -> "
              f"{log['PSNR'][-1]:.3f}  cams={rec.cams[:6]}..")


if __name__ == "__main__":
    main()
