"""Host mirror of the mesh-extraction entry points of the reference's utils/util.py that sit on the SDF path.

`extract_mesh` (utils/util.py:392-425) evaluates `implicit_surface.infer_sdf` on an N^3 lattice in 16 k-point chunks
(host numpy lattice -> device -> host per chunk) and hands the volume to skimage's marching cubes.  Here the sweep is ONE
device call (`ls2fm_sdf_volume`: the lattice is generated in the kernel, the volume stays in HBM); marching cubes and the
PLY writer are third-party packages (scikit-image, plyfile) that are used when installed -- they are not part of the path.
"""
import numpy as np
import torch

from .. import fused


def _lattice(volume_size, N, bound_max, bound_min):
    # utils/util.py:395-409: the step is ALWAYS volume_size / (N - 1) (also when bounds are given), and column c of the
    # points is offset by voxel_grid_origin[2 - c]
    s = float(volume_size)
    origin = [-s / 2.0, -s / 2.0, -s / 2.0]
    sizes = [s, s, s]
    if bound_max is not None:
        sizes = [float(i) - float(j) for i, j in zip(bound_max, bound_min)]
        origin = [float(v) for v in bound_min]
    step = [s / (N - 1)] * 3
    return step, [origin[2], origin[1], origin[0]], origin, sizes


def sdf_volume(implicit_surface, volume_size=2.0, N=512, bound_max=None, bound_min=None, reference_indexing=True):
    """the [N, N, N] float32 volume extract_mesh passes to marching cubes, as a tensor on the field's device.
    reference_indexing=True keeps the reference's lattice (numpy true division makes its first two index columns
    fractional: a slightly sheared grid); False samples the regular lattice those lines were meant to build."""
    step, col_origin, _, _ = _lattice(volume_size, N, bound_max, bound_min)
    return fused.sdf_volume(implicit_surface, N, step, col_origin, reference_indexing=reference_indexing).view(N, N, N)


def lattice_points(volume_size=2.0, N=512, bound_max=None, bound_min=None, first=0, count=None, reference_indexing=True):
    """the lattice of utils/util.py:399-409 as a float32 numpy array [count, 3] (host restatement, for checks)"""
    step, col_origin, _, _ = _lattice(volume_size, N, bound_max, bound_min)
    count = N ** 3 - first if count is None else count
    idx = np.arange(first, first + count, 1).astype(np.int64)
    xyz = np.zeros([count, 3])
    if reference_indexing:
        xyz[:, 2] = idx % N
        xyz[:, 1] = (idx / N) % N
        xyz[:, 0] = ((idx / N) / N) % N
    else:
        xyz[:, 2] = idx % N
        xyz[:, 1] = (idx // N) % N
        xyz[:, 0] = idx // (N * N)
    for c in range(3):
        xyz[:, c] = xyz[:, c] * step[c] + col_origin[c]
    return xyz.astype(np.float32)


def extract_mesh(implicit_surface, log=None, volume_size=2.0, level=0.0, N=512, filepath="./surface.ply",
                 show_progress=True, chunk=16 * 1024, bound_max=None, bound_min=None, extra_info=None):
    """utils/util.py:392-425.  `chunk` and `show_progress` are accepted and unused (one device call)."""
    if extra_info is not None:
        raise NotImplementedError("extract_mesh(extra_info=...): residual fields are outside the fused path")
    vol = sdf_volume(implicit_surface, volume_size, N, bound_max, bound_min)
    _, _, origin, sizes = _lattice(volume_size, N, bound_max, bound_min)
    try:
        import skimage.measure
        import plyfile
    except ImportError as e:          # third-party, not vendored: the volume is the product of this path
        raise ImportError("extract_mesh needs scikit-image and plyfile for marching cubes / PLY output; "
                          "ls2fm.utils.util.sdf_volume returns the volume itself") from e
    verts, faces, _, _ = skimage.measure.marching_cubes(vol.cpu().numpy(), level=level,
                                                        spacing=[float(v) / N for v in sizes])
    pts = verts + np.asarray(origin, dtype=verts.dtype)[None, :]
    vt = np.zeros((len(pts),), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4")])
    vt["x"], vt["y"], vt["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    ft = np.zeros((len(faces),), dtype=[("vertex_indices", "i4", (3,))])
    ft["vertex_indices"] = faces
    plyfile.PlyData([plyfile.PlyElement.describe(vt, "vertex"), plyfile.PlyElement.describe(ft, "face")]).write(filepath)
    if log is not None:
        log.info("saving mesh to %s" % str(filepath))
    return vol


# ------------------------------------------------------------------------------------------------ checkpoints
# The reference's `model.ckpt` (utils/util.py:198-259): {epoch, iter, sdf_func, color_func, cam_info, pts3d_info,
# optim_* / sched_*}.  SDF / RadF here expose the reference's state_dict keys and shapes, so its
# checkpoints load into these classes and checkpoints written here load into the reference.
def save_checkpoint_sfm(opt, model, ep, it, latest=False):
    """utils/util.py:239-259"""
    import os
    import shutil
    os.makedirs("{0}/model".format(opt.output_path), exist_ok=True)
    cams = getattr(model, "camera_set", None)
    pts = getattr(model, "point_set", None)
    checkpoint = dict(
        epoch=ep,
        iter=it,
        sdf_func=model.sdf_func.state_dict(),
        color_func=model.color_func.state_dict(),
        cam_info=cams.get_all_parameters() if cams is not None else None,
        pts3d_info=pts.get_all_parameters() if pts is not None else None,
    )
    for key in model.__dict__:
        if key.split("_")[0] in ["optim", "sched"]:
            checkpoint.update({key: getattr(model, key).state_dict()})
    torch.save(checkpoint, "{0}/model.ckpt".format(opt.output_path))
    if not latest:
        shutil.copy("{0}/model.ckpt".format(opt.output_path),
                    "{0}/model/{1}.ckpt".format(opt.output_path, ep or it))   # ep None: track the iteration instead


def restore_checkpoint_sfm(opt, model, load_name=None, resume=False):
    """utils/util.py:198-218: (epoch, iteration) when resuming, else (None, None)"""
    assert ((load_name is None) == (resume is not False))       # resume: True / False / an epoch or iteration number
    if resume:
        load_name = "{0}/model.ckpt".format(opt.output_path) if resume is True else \
            "{0}/model/{1}.ckpt".format(opt.output_path, resume)
    checkpoint = torch.load(load_name, map_location=opt.device, weights_only=False)
    model.sdf_func.load_state_dict(checkpoint["sdf_func"], False)      # non-strict, like the reference
    model.color_func.load_state_dict(checkpoint["color_func"])
    model.cam_info_reloaded = checkpoint["cam_info"]
    model.pts_info_reloaded = checkpoint["pts3d_info"]
    for key in model.__dict__:
        if key.split("_")[0] in ["optim", "sched"] and key in checkpoint and resume:
            getattr(model, key).load_state_dict(checkpoint[key])
    if resume:
        ep, it = checkpoint["epoch"], checkpoint["iter"]
        if resume is not True:
            assert (resume == (ep or it))
    else:
        ep, it = None, None
    return ep, it
