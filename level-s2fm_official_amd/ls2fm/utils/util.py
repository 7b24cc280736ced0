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
# File format of the reference's `model.ckpt` (written at utils/util.py:239-259, read at :198-218) -- the FORMAT is the
# interface: one torch-pickled dict with the keys below; `<output_path>/model.ckpt` is the latest state and
# `<output_path>/model/<tag>.ckpt` a numbered snapshot.  SDF / RadF here expose the reference's state_dict keys and
# shapes, so files written by either side load into the other.
from ..checkpoint import (latest_path, snapshot_path, collect_checkpoint, apply_checkpoint,  # noqa: E402
                          write_checkpoint, read_checkpoint)


def save_checkpoint_sfm(opt, model, ep, it, latest=False):
    """same call as utils/util.py:239: writes the latest file and, unless `latest`, a snapshot tagged ep (or it)"""
    payload = collect_checkpoint(model, epoch=ep, iteration=it)
    write_checkpoint(payload, opt.output_path, tag=None if latest else (ep or it))


def restore_checkpoint_sfm(opt, model, load_name=None, resume=False):
    """same call as utils/util.py:198: -> (epoch, iteration) when resuming, (None, None) when only loading weights.
    Exactly one of `load_name` (a file) and `resume` (True = latest, or a snapshot tag) is given."""
    if (load_name is None) != (resume is not False):
        raise AssertionError("give either load_name or resume")
    if resume is True:
        load_name = latest_path(opt.output_path)
    elif resume:
        load_name = snapshot_path(opt.output_path, resume)
    payload = read_checkpoint(load_name, opt.device)
    apply_checkpoint(model, payload, with_training_state=bool(resume))
    if not resume:
        return None, None
    where = payload["epoch"], payload["iter"]
    if resume is not True and resume != (where[0] or where[1]):
        raise AssertionError(f"snapshot {resume} holds epoch {where[0]} / iteration {where[1]}")
    return where
