"""ls2fm -- MI355X-native SDF ray-marching / volumetric rendering hot path of Level-S2fM.

Host side is PyTorch-ROCm plumbing (device memory, streams, autograd, torch.distributed); the
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of include/ls2fm.h.
The class surface mirrors the reference (models/SDF.py, models/RadF.py, models/Renderer.py,
models/base.py, utils/custom_functions.py) so its pipelines can call it unchanged:

    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
"""
from . import _lib, hashgrid, ops, options  # noqa: F401

__all__ = ["_lib", "hashgrid", "ops", "options"]
