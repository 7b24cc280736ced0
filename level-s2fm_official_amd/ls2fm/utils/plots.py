"""SDF sweeps behind the reference's mesh export `utils/plots.py:get_surface_high_res_mesh` (:140-222; called from
pipelines/LevelS2fM.py:228 with `self.sdf_func.infer_sdf`, resolution 512): a 100^3 uniform lattice for a coarse mesh,
then a resolution^3-class lattice aligned with the principal axes of that mesh.

The reference walks both lattices in 100 000-point chunks, each a device evaluation + a device->host copy.  Here a
lattice is built on the device from its three 1-D axes (formed on the host in fp64 exactly as the reference's numpy
lines do, then rounded to fp32 like its `dtype=torch.float`), optionally rotated on the device, and evaluated by the fused
no-graph SDF kernel in one pass (`ls2fm.fused.sdf_eval`; chunks of 2^25 points only bound the scratch memory); the volume
stays in HBM until marching cubes wants it.  Marching cubes, connected components, surface sampling and PLY export are
third-party (scikit-image, trimesh) and are used when installed -- they are not part of the path.
"""
from __future__ import annotations

import numpy as np
import torch

_CHUNK = 1 << 25


def uniform_axes(resolution, grid_boundary=(-2.0, 2.0)):
    """utils/plots.py:325-336 (get_grid_uniform): the same linspace on every axis"""
    x = np.linspace(grid_boundary[0], grid_boundary[1], resolution)
    return [x, x, x]


def fitted_axes(points, resolution, input_min=None, input_max=None, eps=0.1):
    """utils/plots.py:338-370 (get_grid): `resolution` steps along the SHORTEST side of the points' bounding box, the same
    step along the other two (np.arange, so their counts follow from the box) -> ([x, y, z], shortest length, its index)"""
    if input_min is None or input_max is None:
        pts = torch.as_tensor(points).detach().cpu()
        input_min = pts.min(dim=0).values.numpy()
        input_max = pts.max(dim=0).values.numpy()
    input_min, input_max = np.asarray(input_min), np.asarray(input_max)
    k = int(np.argmin(input_max - input_min))
    short = np.linspace(input_min[k] - eps, input_max[k] + eps, resolution)
    length = np.max(short) - np.min(short)
    step = length / (short.shape[0] - 1)
    axes = [short if a == k else np.arange(input_min[a] - eps, input_max[a] + step + eps, step) for a in range(3)]
    return axes, length, k


def lattice_on_device(axes, device, first=0, count=None):
    """rows first..first+count of np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T for xx, yy, zz = np.meshgrid(x, y, z)
    (numpy's default 'xy' indexing: the flat index runs over (y, x, z), z fastest), as float32 [count, 3]"""
    x, y, z = [torch.from_numpy(np.asarray(a)).to(torch.float32).to(device) for a in axes]
    nx, nz = x.numel(), z.numel()
    total = y.numel() * nx * nz
    count = total - first if count is None else count
    idx = torch.arange(first, first + count, device=device, dtype=torch.int64)
    iz = idx % nz
    ix = (idx // nz) % nx
    iy = idx // (nz * nx)
    return torch.stack([x[ix], y[iy], z[iz]], dim=1)


def sdf_on_lattice(sdf, axes, rotation=None, offset=None, device=None):
    """`sdf` (a callable [M,3] -> [M,1], e.g. `SDF.infer_sdf`) on the meshgrid of `axes`, flat float32 [ny*nx*nz] on the
    device; with `rotation` [3,3] / `offset` [3] the lattice points are first mapped p -> rotation^T p + offset
    (utils/plots.py:190-193).  Runs without a graph: on an `ls2fm` SDF field that is the fused kernel."""
    owner = getattr(sdf, "__self__", None)
    if device is None:
        device = next(owner.parameters()).device if isinstance(owner, torch.nn.Module) else torch.device("cuda")
    total = int(np.prod([len(a) for a in axes]))
    out = torch.empty(total, device=device, dtype=torch.float32)
    rot = None if rotation is None else torch.as_tensor(rotation, dtype=torch.float32, device=device)
    off = None if offset is None else torch.as_tensor(offset, dtype=torch.float32, device=device)
    with torch.no_grad():
        for first in range(0, total, _CHUNK):
            n = min(_CHUNK, total - first)
            pts = lattice_on_device(axes, device, first, n)
            if rot is not None:
                pts = pts @ rot                     # row form of rot^T p
            if off is not None:
                pts = pts + off
            out[first:first + n] = sdf(pts.contiguous()).reshape(-1)
    return out


def _principal_frame(points):
    """utils/plots.py:169-177: centroid and eigenvectors (rows) of the scatter matrix of the sampled surface points,
    made right-handed by swapping the last two axes"""
    mean = points.mean(dim=0)
    centred = points - mean
    cov = centred.t() @ centred
    vecs = torch.view_as_real(torch.linalg.eig(cov)[1].t())[:, :, 0]
    if torch.det(vecs) < 0:
        vecs = torch.tensor([[1.0, 0, 0], [0, 0, 1.0], [0, 1.0, 0]], device=points.device) @ vecs
    return mean, vecs


def get_surface_high_res_mesh(sdf, resolution=100, grid_boundary=(-2.0, 2.0), level=0, take_components=True, path=None):
    """utils/plots.py:140-222, same arguments.  The two SDF sweeps are device passes (see module docstring); everything
    that turns the volumes into a mesh needs scikit-image and trimesh."""
    try:
        from skimage import measure
        import trimesh
    except ImportError as e:
        raise ImportError("get_surface_high_res_mesh needs scikit-image and trimesh for marching cubes / mesh handling; "
                          "ls2fm.utils.plots.sdf_on_lattice returns the volumes themselves") from e
    axes = uniform_axes(100, grid_boundary)
    z = sdf_on_lattice(sdf, axes)
    device = z.device
    sp = axes[0][2] - axes[0][1]
    vol = z.view(len(axes[1]), len(axes[0]), len(axes[2])).permute(1, 0, 2).cpu().numpy()
    verts, faces, normals, _ = measure.marching_cubes(volume=vol, level=level, spacing=(sp, sp, sp))
    verts = verts + np.array([axes[0][0], axes[1][0], axes[2][0]])
    low = trimesh.Trimesh(verts, faces, normals)
    if take_components:
        parts = low.split(only_watertight=False)
        low = parts[int(np.argmax([c.area for c in parts]))]
    cloud = torch.from_numpy(trimesh.sample.sample_surface(low, 10000)[0]).float().to(device)
    mean, vecs = _principal_frame(cloud)
    aligned = (cloud - mean) @ vecs.t()
    axes2, _, _ = fitted_axes(aligned.cpu(), resolution)
    z2 = sdf_on_lattice(sdf, axes2, rotation=vecs, offset=mean, device=device)
    mesh = None
    if not (float(z2.min()) > level or float(z2.max()) < level):
        sp2 = axes2[0][2] - axes2[0][1]
        vol2 = z2.view(len(axes2[1]), len(axes2[0]), len(axes2[2])).permute(1, 0, 2).cpu().numpy()
        verts, faces, normals, _ = measure.marching_cubes(volume=vol2, level=level, spacing=(sp2, sp2, sp2))
        first = lattice_on_device(axes2, device, 0, 1) @ vecs + mean
        verts = (torch.from_numpy(verts).float().to(device) @ vecs + first[0]).cpu().numpy()
        mesh = trimesh.Trimesh(verts, faces, normals)
    if path is not None and mesh is not None:
        with open(path, "wb+") as f:
            f.write(trimesh.exchange.ply.export_ply(mesh, encoding="ascii"))
    return mesh
