"""Reading and writing the reference's `model.ckpt` files (format defined by utils/util.py:198-259 of the reference).

The file is one torch-pickled dict:

    epoch, iter                 progress counters (either may be None)
    sdf_func, color_func        state_dicts of the SDF / radiance fields (keys and shapes: SURVEY App. E)
    cam_info, pts3d_info        whatever the pipeline's camera / point sets report through get_all_parameters()
    optim_*, sched_*            state_dicts of every attribute of the pipeline object whose name starts that way

Layout on disk: `<output_path>/model.ckpt` (latest) and `<output_path>/model/<tag>.ckpt` (snapshots, tag = epoch or
iteration).  The field classes of this package keep the reference's parameter names, so files travel both ways.
"""
from __future__ import annotations

import shutil
from pathlib import Path

import torch

FIELD_KEYS = (("sdf_func", "sdf_func"), ("color_func", "color_func"))       # payload key, pipeline attribute
SET_KEYS = (("cam_info", "camera_set"), ("pts3d_info", "point_set"))
TRAINING_PREFIXES = ("optim", "sched")


def latest_path(output_path) -> Path:
    return Path(output_path) / "model.ckpt"


def snapshot_path(output_path, tag) -> Path:
    return Path(output_path) / "model" / f"{tag}.ckpt"


def _training_attrs(model):
    """names of the optimizer / scheduler attributes of a pipeline object (`optim_sdf`, `sched_color`, ...)"""
    return [name for name in vars(model) if name.partition("_")[0] in TRAINING_PREFIXES]


def collect_checkpoint(model, epoch=None, iteration=None) -> dict:
    payload = {"epoch": epoch, "iter": iteration}
    for key, attr in FIELD_KEYS:
        payload[key] = getattr(model, attr).state_dict()
    for key, attr in SET_KEYS:
        holder = getattr(model, attr, None)
        payload[key] = None if holder is None else holder.get_all_parameters()
    for name in _training_attrs(model):
        payload[name] = getattr(model, name).state_dict()
    return payload


def write_checkpoint(payload: dict, output_path, tag=None) -> Path:
    """latest file always; with a tag also the numbered snapshot (a copy of the same bytes)"""
    newest = latest_path(output_path)
    (newest.parent / "model").mkdir(parents=True, exist_ok=True)
    torch.save(payload, str(newest))
    if tag is not None:
        shutil.copyfile(newest, snapshot_path(output_path, tag))
    return newest


def read_checkpoint(path, device="cpu") -> dict:
    return torch.load(str(path), map_location=device, weights_only=False)


def apply_checkpoint(model, payload: dict, with_training_state=False) -> None:
    """fields always (the SDF field non-strictly, as the reference loads it: older files lack some keys); optimizer /
    scheduler state only when resuming a run"""
    model.sdf_func.load_state_dict(payload["sdf_func"], strict=False)
    model.color_func.load_state_dict(payload["color_func"])
    model.cam_info_reloaded = payload["cam_info"]
    model.pts_info_reloaded = payload["pts3d_info"]
    if with_training_state:
        for name in _training_attrs(model):
            if name in payload:
                getattr(model, name).load_state_dict(payload[name])
