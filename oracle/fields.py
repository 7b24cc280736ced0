"""Oracle: SDF field, radiance field, renderer and sphere tracing of Level-S2fM.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional restatement (plain torch
CPU ops + autograd) of the reference's in-tree Python:

  hash_embed          models/base.py:23-40    (Embedder_Hash.forward)
  fourier_embed       models/base.py:75-97    (Embedder_Fourier.forward, kwargs :143-151)
  geometry_mlp        models/base.py:206-217  (Geometry.forward; weight_norm :200; Softplus(100,20) :203)
  radiance_mlp        models/base.py:249-261  (Radiance.forward; NOTE no ReLU ever fires, SURVEY C-1)
  infer_sdf           models/SDF.py:55-78
  forward_ab          models/SDF.py:80-82
  sdf_to_sigma        models/SDF.py:84-87
  sdf_gradient        models/SDF.py:102-114
  get_surface_pts     models/SDF.py:95-100
  sphere_tracing      models/SDF.py:116-226
  geometry_feat       models/RadF.py:66-76
  sample_depth        models/Renderer.py:118-127
  composite           models/Renderer.py:33-49
  render              models/Renderer.py:51-116 (+ volsdf_sampling default branch :169-185,
                      utils/camera.py:262-266 for p = c + d*t)

Parameters are plain dicts keyed exactly like the reference's ``state_dict()``
 so weights captured from the imported reference load directly.
PINNED against golden vectors generated from the imported reference
(tests/golden/make_golden.py -> tests/test_oracle_vs_golden.py), except for the two
third-party ops it calls (oracle.hashgrid, oracle.ray_aabb: parity unpinned).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import hashgrid, ray_aabb

State = Dict[str, torch.Tensor]


@dataclass
class PathConfig:
    """The option keys the hot path reads (SURVEY.md section 5 'config / flags'; values are the
    reference's defaults from options/LevelS2fM.yaml unless a dataset yaml overrides them)."""

    bound_min: Sequence[float] = (-1.0, -1.0, -1.0)
    bound_max: Sequence[float] = (1.0, 1.0, 1.0)
    inside: bool = True                 # data.inside  -> sign of the sdf (SDF.py:66-71)
    bg_sdf: Optional[bool] = None       # data.bg_sdf  (unset in every shipped config)
    bg_rad: float = 2.0
    bgcolor: Sequence[float] = (0.0, 0.0, 0.0)
    scale_mlp: float = 1.0              # SDF.NN_Init.scale_mlp
    bias: float = 1.0                   # SDF.NN_Init.bias (geometric init sphere radius)
    rescale: float = 1.0                # SDF.VolSDF.rescale
    beta_speed: float = 1.0
    beta_init: float = 0.05
    sdf_threshold: float = 1e-3
    iters_max_st: int = 20
    sample_intvs: int = 128
    res: int = 100                      # opt.Res (finish-mask threshold of sphere tracing)
    dual_field: bool = False
    # hash grid (options/config_hash_sdf.json) -- per_level_scale is derived, base.py:128-129
    n_levels: int = 16
    n_features: int = 2
    log2_hashmap_size: int = 19
    base_resolution: int = 16
    hidden: int = 64                    # SDF.arch.layers = [null, 64, 16]
    feat_dim: int = 16
    rad_hidden: Sequence[int] = (64, 64)  # RadF.arch.layers = [null, 64, 64, 3]

    def table(self) -> hashgrid.LevelTable:
        b = hashgrid.reference_per_level_scale(self.bound_min[0], self.bound_max[0], self.n_levels,
                                               self.base_resolution)
        return hashgrid.make_level_table(self.n_levels, self.n_features, self.log2_hashmap_size,
                                         self.base_resolution, b)

    def bounds(self, dtype=torch.float32):
        return (torch.tensor(self.bound_min, dtype=dtype), torch.tensor(self.bound_max, dtype=dtype))


DATASETS = {
    # options/DTU.yaml:3-15, ETH3D.yaml:2-13, bmvs.yaml:5-15, Scannet.yaml:3-16
    "DTU": dict(bound_min=(-1, -1, -1), bound_max=(1, 1, 1), inside=True, scale_mlp=1.0, bias=0.5,
                iters_max_st=10, bgcolor=(0, 0, 0)),
    "ETH3D": dict(bound_min=(-5, -5, -5), bound_max=(5, 5, 5), inside=False, scale_mlp=5.0, bias=2.5,
                  iters_max_st=20, bgcolor=(0, 0, 0)),
    "BlendedMVS": dict(bound_min=(-2, -2, -2), bound_max=(2, 2, 2), inside=True, scale_mlp=3.0, bias=1.0,
                       iters_max_st=20, bgcolor=(1, 1, 1)),
    "scannet": dict(bound_min=(-4, -4, -4), bound_max=(4, 4, 4), inside=False, scale_mlp=1.0, bias=2.0,
                    iters_max_st=10, bgcolor=(0, 0, 0)),
}


def dataset_config(name: str, **overrides) -> PathConfig:
    kw = dict(DATASETS[name])
    kw.update(overrides)
    return PathConfig(**kw)


# ----------------------------------------------------------------------------- parameters
def _wn_pack(w: torch.Tensor):
    """legacy torch.nn.utils.weight_norm parametrisation (dim=0): g = row norm, v = w."""
    return w.norm(dim=1, keepdim=True).clone(), w.clone()


def init_sdf_state(cfg: PathConfig, gen: torch.Generator, prefix_mlp: str = "SDF_MLP") -> State:
    """Geometric initialisation (models/base.py:184-199) + table U(-1e-4,1e-4) + beta (SDF.py:28-32)."""
    table = cfg.table()
    in_dim = 3 + table.n_output_dims
    sd: State = {}
    sd["embed_fn.embedder_obj.params"] = (torch.rand(table.n_params, generator=gen) * 2 - 1) * 1e-4
    w0 = torch.zeros(cfg.hidden, in_dim)
    w0[:, :3] = torch.randn(cfg.hidden, 3, generator=gen) * (math.sqrt(2) / math.sqrt(cfg.hidden))
    b0 = torch.zeros(cfg.hidden)
    w1 = torch.randn(cfg.feat_dim + 1, cfg.hidden, generator=gen) * 1e-4 + math.sqrt(math.pi) / math.sqrt(cfg.hidden)
    b1 = torch.full((cfg.feat_dim + 1,), -float(cfg.bias))
    for li, (w, b) in enumerate(((w0, b0), (w1, b1))):
        g, v = _wn_pack(w)
        sd[f"{prefix_mlp}.mlp.{li}.bias"] = b
        sd[f"{prefix_mlp}.mlp.{li}.weight_g"] = g
        sd[f"{prefix_mlp}.mlp.{li}.weight_v"] = v
    if prefix_mlp == "SDF_MLP":
        sd["beta"] = torch.tensor([math.log(cfg.beta_init) / cfg.beta_speed], dtype=torch.float32)
    return sd


def rad_input_dim(cfg: PathConfig) -> int:
    # 3 point + 3 normal + 27 view embedding + geo feature(s)   (models/RadF.py:54-58)
    return 3 + 3 + 27 + cfg.feat_dim * (2 if cfg.dual_field else 1)


def init_rad_state(cfg: PathConfig, gen: torch.Generator) -> State:
    sd: State = {}
    if cfg.dual_field:
        geo = init_sdf_state(cfg, gen, prefix_mlp="Geo_enc")
        sd.update(geo)
    dims = [rad_input_dim(cfg), *cfg.rad_hidden, 3]
    for li, (k_in, k_out) in enumerate(zip(dims[:-1], dims[1:])):
        bound = 1.0 / math.sqrt(k_in)        # torch.nn.Linear default init
        w = (torch.rand(k_out, k_in, generator=gen) * 2 - 1) * bound
        b = (torch.rand(k_out, generator=gen) * 2 - 1) * bound
        g, v = _wn_pack(w)
        sd[f"Rad_dec.mlp_radiance.{li}.bias"] = b
        sd[f"Rad_dec.mlp_radiance.{li}.weight_g"] = g
        sd[f"Rad_dec.mlp_radiance.{li}.weight_v"] = v
    return sd


def randomize_state(sd: State, gen: torch.Generator, table_amp: float = 0.1, w_std: float = 0.05) -> None:
    """Make the hash path live: at geometric init W0[:,3:] == 0 so every table gradient is exactly
    zero.  Tables -> U(-amp, amp); first-layer hash columns -> N(0, w_std)."""
    for k in list(sd.keys()):
        if k.endswith("embedder_obj.params"):
            sd[k] = (torch.rand(sd[k].shape, generator=gen) * 2 - 1) * table_amp
        if k.endswith("mlp.0.weight_v") and "Rad_dec" not in k:
            v = sd[k].clone()
            v[:, 3:] = torch.randn(v[:, 3:].shape, generator=gen) * w_std
            sd[k] = v
            sd[k.replace("weight_v", "weight_g")] = v.norm(dim=1, keepdim=True)


def to_dtype(sd: State, dtype, requires_grad: bool = False) -> State:
    return {k: v.detach().to(dtype).clone().requires_grad_(requires_grad) for k, v in sd.items()}


# ----------------------------------------------------------------------------- building blocks
def eff_weight(sd: State, prefix: str) -> torch.Tensor:
    """W = g * v / ||v||_row  (legacy weight_norm, dim=0)."""
    return torch._weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"], 0)


def softplus100(a: torch.Tensor) -> torch.Tensor:
    return F.softplus(a, beta=100, threshold=20)


def hash_embed(xyz: torch.Tensor, params: torch.Tensor, cfg: PathConfig, table) -> torch.Tensor:
    bmin, bmax = cfg.bounds(xyz.dtype)
    x = (xyz - bmin) / (bmax - bmin)
    enc = hashgrid.encode(x.reshape(-1, 3), params, table)
    return torch.cat([xyz / cfg.rescale, enc.view(*xyz.shape[:-1], -1)], dim=-1)


def fourier_embed(d: torch.Tensor) -> torch.Tensor:
    out = [d]
    for freq in (1.0, 2.0, 4.0, 8.0):       # 2 ** linspace(0, 3, 4)
        out.append(torch.sin(d * freq))
        out.append(torch.cos(d * freq))
    return torch.cat(out, dim=-1)


def geometry_mlp(enc: torch.Tensor, sd: State, prefix: str) -> torch.Tensor:
    h = softplus100(F.linear(enc, eff_weight(sd, f"{prefix}.mlp.0"), sd[f"{prefix}.mlp.0.bias"]))
    return F.linear(h, eff_weight(sd, f"{prefix}.mlp.1"), sd[f"{prefix}.mlp.1.bias"])


def radiance_mlp(x: torch.Tensor, sd: State, n_layers: int = 3) -> torch.Tensor:
    for li in range(n_layers):
        x = F.linear(x, eff_weight(sd, f"Rad_dec.mlp_radiance.{li}"), sd[f"Rad_dec.mlp_radiance.{li}.bias"])
    return torch.sigmoid(x)


def infer_sdf(xyz: torch.Tensor, sdf_sd: State, cfg: PathConfig, table, mode: str = "ret_sdf"):
    enc = hash_embed(xyz, sdf_sd["embed_fn.embedder_obj.params"], cfg, table)
    feat = geometry_mlp(enc, sdf_sd, "SDF_MLP")
    if cfg.inside:
        sdf = feat[..., :1] / cfg.scale_mlp
        if cfg.bg_sdf is True:
            sdf = torch.min(sdf, cfg.bg_rad - xyz.norm(dim=-1, keepdim=True))
    else:
        sdf = -feat[..., :1] / cfg.scale_mlp
    if mode == "ret_sdf":
        return sdf
    if mode == "ret_feat":
        return feat
    return sdf, feat


def geometry_feat(xyz: torch.Tensor, rad_sd: State, cfg: PathConfig, table) -> torch.Tensor:
    enc = hash_embed(xyz, rad_sd["embed_fn.embedder_obj.params"], cfg, table)
    return geometry_mlp(enc, rad_sd, "Geo_enc")


def forward_ab(sdf_sd: State, cfg: PathConfig):
    beta = torch.exp(sdf_sd["beta"] * cfg.beta_speed)
    return 1.0 / beta, beta


def sdf_to_sigma(sdf: torch.Tensor, alpha, beta) -> torch.Tensor:
    e = 0.5 * torch.exp(-torch.abs(sdf) / beta)
    return alpha * torch.where(sdf >= 0, e, 1 - e)


def sdf_gradient(p: torch.Tensor, sdf_sd: State, cfg: PathConfig, table) -> torch.Tensor:
    """d sdf / d p with the graph kept (callers differentiate through it)."""
    with torch.enable_grad():
        if not p.requires_grad:
            p.requires_grad_(True)
        y = infer_sdf(p, sdf_sd, cfg, table, "ret_sdf")
        (g,) = torch.autograd.grad(y, p, torch.ones_like(y), create_graph=True, retain_graph=True)
    return g


def get_surface_pts(pts: torch.Tensor, sdf_sd: State, cfg: PathConfig, table):
    sdf = infer_sdf(pts.detach(), sdf_sd, cfg, table, "ret_sdf")
    n = sdf_gradient(pts, sdf_sd, cfg, table)
    n_len = torch.norm(n, dim=-1, keepdim=True)
    return pts - n / n_len.detach() * sdf, n_len


# ----------------------------------------------------------------------------- renderer
def scene_box(cfg: PathConfig, dtype=torch.float32):
    bmin, bmax = cfg.bounds(dtype)
    return (bmax + bmin) / 2, (bmax - bmin) / 2


def sample_depth(n_samples: int, near: torch.Tensor, far: torch.Tensor) -> torch.Tensor:
    """near, far [B,R,1] -> mid-point samples [B,R,N,1]."""
    steps = 0.5 + torch.arange(n_samples, dtype=torch.float32)[None, None, :, None].to(near.dtype)
    return steps / n_samples * (far[..., None, :] - near[..., None, :]) + near[..., None, :]


def composite(ray: torch.Tensor, rgb_s: torch.Tensor, sigma_s: torch.Tensor, t_s: torch.Tensor):
    """ray [B,R,3], rgb_s [B,R,N,3], sigma_s [B,R,N], t_s [B,R,N,1] -> rgb [B,R,3], prob [B,R,N-1,1]."""
    ray_len = ray.norm(dim=-1, keepdim=True)
    dist = (t_s[..., 1:, 0] - t_s[..., :-1, 0]) * ray_len
    sd = sigma_s[..., :-1] * dist
    alpha = 1 - torch.exp(-sd)
    acc = torch.cat([torch.zeros_like(sd[..., :1]), sd], dim=2).cumsum(dim=2)
    trans = torch.exp(-acc)[..., :-1]
    prob = (trans * alpha)[..., None]
    return (rgb_s[..., :-1, :] * prob).sum(dim=2), prob


def render(cfg: PathConfig, center: torch.Tensor, ray: torch.Tensor, sdf_sd: State, rad_sd: State,
           table=None, rad_table=None) -> Dict[str, torch.Tensor]:
    """Renderer.forward: center, ray [B,R,3] -> dict(rgb, sdfs_volume, normals, depth_mlp, normal_mlp)."""
    table = table or cfg.table()
    rad_table = rad_table or table
    dtype = center.dtype
    box_c, box_h = scene_box(cfg, torch.float32)
    near, far = ray_aabb.near_far(center.reshape(-1, 3), ray.reshape(-1, 3), box_c, box_h)
    near = near.view(*center.shape[:2], 1).to(dtype)
    far = far.view(*center.shape[:2], 1).to(dtype)
    t = sample_depth(cfg.sample_intvs, near, far)                       # [B,R,N,1]
    p = center[:, :, None] + ray[:, :, None] * t                         # [B,R,N,3]
    alpha, beta = forward_ab(sdf_sd, cfg)
    sdfs, feats = infer_sdf(p, sdf_sd, cfg, table, "ret_all")
    normals = sdf_gradient(p, sdf_sd, cfg, table)
    ray_enc = fourier_embed(ray[..., None, :].expand_as(p))
    if cfg.dual_field:
        geo = torch.cat([feats[..., 1:], geometry_feat(p, rad_sd, cfg, rad_table)[..., 1:]], dim=-1)
    else:
        geo = feats[..., 1:]
    rgbs = radiance_mlp(torch.cat([p, normals, ray_enc, geo], dim=-1), rad_sd)
    return render_tail(cfg, ray, t, sdfs, normals, rgbs, alpha, beta)


def render_tail(cfg: PathConfig, ray, t, sdfs, normals, rgbs, alpha, beta) -> Dict[str, torch.Tensor]:
    """the part of Renderer.forward after the per-sample field evaluations: sigma, composite, background / depth / normal
    epilogue (Renderer.py:80-107)"""
    sigma = sdf_to_sigma(sdfs, alpha, beta)
    rgb, prob = composite(ray, rgbs, sigma.squeeze(-1), t)
    opacity = prob.sum(dim=2)
    bg = torch.tensor(cfg.bgcolor, dtype=sdfs.dtype)
    rgb = rgb + (1 - opacity) * bg
    depth = (t[..., :-1, :] * prob).sum(dim=2) + (1 - opacity) * t[..., -1, :]
    normal = (normals[..., :-1, :] * prob).sum(dim=2) + (1 - opacity) * normals[..., -1, :]
    return {"rgb": rgb, "sdfs_volume": sdfs, "normals": normals, "depth_mlp": depth, "normal_mlp": normal}


def beta_gradient_exact_sum(cfg: PathConfig, center, ray, sdf_sd: State, rad_sd: State, loss_fn, with_condition=False):
    """d loss / d beta of the FLOAT32 computation with the ill-conditioned part done exactly (test adjudication; no reference
    counterpart).  d/d beta is one scalar summing terms of both signs over every sample (cancellation ~1e3), so an fp32
    autograd value -- the reference's included -- carries summation noise of ~1e-4.  Running the whole oracle in fp64 does
    not give "the true value" of what the fp32 path computes either: the hash-grid weights frac(scale * x + 0.5) lose up to
    ~5e-4 absolute at the finest levels in fp32, so the fp64 FIELD is a slightly different field.  Here the per-sample field
    outputs (sdf, normal, colour: what beta does not touch) are evaluated in fp32 exactly as the path does, and everything
    beta enters -- sigma, composite, epilogue, loss -- is then carried out in fp64 on those values."""
    table = cfg.table()
    with torch.no_grad():
        box_c, box_h = scene_box(cfg, torch.float32)
        near, far = ray_aabb.near_far(center.reshape(-1, 3), ray.reshape(-1, 3), box_c, box_h)
        near = near.view(*center.shape[:2], 1)
        far = far.view(*center.shape[:2], 1)
        t = sample_depth(cfg.sample_intvs, near, far)
        p = center[:, :, None] + ray[:, :, None] * t
        sdfs, feats = infer_sdf(p, sdf_sd, cfg, table, "ret_all")
    normals = sdf_gradient(p.clone(), sdf_sd, cfg, table).detach()
    with torch.no_grad():
        ray_enc = fourier_embed(ray[..., None, :].expand_as(p))
        geo = feats[..., 1:]
        if cfg.dual_field:
            geo = torch.cat([geo, geometry_feat(p, rad_sd, cfg, table)[..., 1:]], dim=-1)
        rgbs = radiance_mlp(torch.cat([p, normals, ray_enc, geo], dim=-1), rad_sd)
    b64 = sdf_sd["beta"].detach().double().clone().requires_grad_(True)
    alpha, beta = forward_ab({"beta": b64}, cfg)
    if not with_condition:
        out = render_tail(cfg, ray.detach().double(), t.double(), sdfs.double(), normals.double(), rgbs.double(), alpha, beta)
        loss_fn(out).backward()
        return b64.grad
    # the same with sigma as an explicit leaf, to see the summands
    s64 = sdfs.double()
    sigma = sdf_to_sigma(s64, alpha, beta)
    sig_leaf = sigma.detach().clone().requires_grad_(True)
    rgb, prob = composite(ray.detach().double(), rgbs.double(), sig_leaf.squeeze(-1), t.double())
    opacity = prob.sum(dim=2)
    bg = torch.tensor(cfg.bgcolor, dtype=torch.float64)
    n64, t64 = normals.double(), t.double()
    out = {"rgb": rgb + (1 - opacity) * bg, "sdfs_volume": s64, "normals": n64,
           "depth_mlp": (t64[..., :-1, :] * prob).sum(dim=2) + (1 - opacity) * t64[..., -1, :],
           "normal_mlp": (n64[..., :-1, :] * prob).sum(dim=2) + (1 - opacity) * n64[..., -1, :]}
    loss_fn(out).backward()
    g_sigma = sig_leaf.grad                                          # dL / dsigma_i
    with torch.no_grad():
        b, a = beta.detach(), alpha.detach()
        e = 0.5 * torch.exp(-s64.abs() / b)
        psi = torch.where(s64 >= 0, e, 1 - e)
        dpsi_db = torch.where(s64 >= 0, e, -e) * s64.abs() / (b * b)
        dsig_dbeta = -a / b * psi + a * dpsi_db                      # d sigma / d beta with alpha = 1 / beta
        terms = g_sigma * dsig_dbeta * b * cfg.beta_speed            # chain to the log-space parameter (SDF.py:28-32)
        total = terms.sum()
    return total.reshape(1), float(terms.abs().sum() / (total.abs() + 1e-300))


# ----------------------------------------------------------------------------- sphere tracing
def sphere_tracing(cfg: PathConfig, ray0: torch.Tensor, ray_dir: torch.Tensor, sdf_sd: State, table=None,
                   rng: bool = True, loop_field=None, details: Optional[dict] = None):
    """Bidirectional sphere tracing with a differentiable depth (SDF.py:116-226).

    ray0, ray_dir [B,R,3] -> d_pred [B,R], sdf_last [B*R], sampled_pts [1,<=4096+B*R,3] (None when
    ``rng`` is False: that output is RNG-dependent), finish_mask [B*R,1] bool; plus the trip count.

    Test hooks (no reference counterpart): ``loop_field(p [M,3]) -> sdf [M]`` replaces the field evaluation INSIDE the
    no-grad root-find loop only (the loop's own arithmetic and masks stay the oracle's: a test can feed it the device's
    field evaluation and so pin the loop semantics trip by trip without the round-off amplification of ``t += sdf``);
    ``details`` (a dict) receives the track [R,K,3], the far-end distance after every trip [R,K+1] and near / far."""
    table = table or cfg.table()
    if loop_field is None:
        def loop_field(q):
            return infer_sdf(q, sdf_sd, cfg, table)[:, 0]
    box_c, box_h = scene_box(cfg, torch.float32)
    shape2 = ray_dir.shape[:2]
    o = ray0.reshape(-1, 3)
    d = ray_dir.reshape(-1, 3)
    near, far = ray_aabb.near_far(o, d, box_c, box_h)
    near = near.to(ray0.dtype)
    far = far.to(ray0.dtype)
    thr = cfg.sdf_threshold
    t_s, t_e = near.clone(), far.clone()
    with torch.no_grad():
        p_s = (o + t_s[:, None] * d)
        p_e = (o + t_e[:, None] * d)
        sdf_s = loop_field(p_s).clone()
        sdf_e = loop_field(p_e).clone()
        unf_s = unf_e = None
        track = []
        t_e_hist = [t_e]
        trips = 0
        while True:
            # (1) converged values are zeroed in place on the persistent arrays
            sdf_s = torch.where(sdf_s.abs() <= thr, torch.zeros_like(sdf_s), sdf_s)
            sdf_e = torch.where(sdf_e.abs() <= thr, torch.zeros_like(sdf_e), sdf_e)
            # (2) masks
            m_s, m_e = sdf_s.abs() > thr, sdf_e.abs() > thr
            unf_s = m_s if unf_s is None else unf_s & m_s
            unf_e = m_e if unf_e is None else unf_e & m_e
            # (3) global break: only the *start* mask is tested
            if int(unf_s.sum()) == 0 or trips == cfg.iters_max_st:
                break
            trips += 1
            # (4) every ray steps by its (possibly stale / zeroed) sdf; both ends step with '+'
            t_s = t_s + 1.0 * sdf_s
            t_e = t_e + 1.0 * sdf_e
            t_s = torch.where(t_s > far, far, t_s)
            t_e = torch.where(t_e > far, far, t_e)
            # (5) the pre-update start point goes on the track
            track.append(p_s)
            p_s = (o + t_s[:, None] * d)
            p_e = (o + t_e[:, None] * d)
            # (6) refresh sdf only where still unfinished
            if int(unf_s.sum()) > 0:
                sdf_s = sdf_s.clone()
                sdf_s[unf_s] = loop_field(p_s[unf_s])
            if int(unf_e.sum()) > 0:
                sdf_e = sdf_e.clone()
                sdf_e[unf_e] = loop_field(p_e[unf_e])
            # (7) rays whose ends crossed drop out (and keep their stale sdf)
            cross = t_s < t_e
            unf_s = unf_s & cross
            unf_e = unf_e & cross
            t_e_hist.append(t_e)
        if not track:
            track = [p_s]
        pts_tracks = torch.stack([q.detach() for q in track], dim=1)      # [R,K,3]
        if details is not None:
            details.update(track=pts_tracks, t_end=torch.stack(t_e_hist, dim=1), near=near, far=far)
    sdf_tracks = infer_sdf(pts_tracks, sdf_sd, cfg, table)                 # grad-enabled  [R,K,1]
    d_pred = sdf_tracks.sum(dim=-2).view(*shape2) + near.view(*shape2)
    d_pred = torch.where(d_pred > far.view(*shape2), far.view(*shape2), d_pred)
    bmin, bmax = cfg.bounds(ray0.dtype)
    finish = sdf_tracks[:, -1, :].abs() < (bmax[0] - bmin[0]) / 10 / cfg.res
    sampled = None
    if rng:
        u = torch.rand_like(d_pred)
        d_up = torch.where(1.5 * t_e.view(*shape2) > far.view(*shape2), far.view(*shape2), 1.5 * t_e.view(*shape2))
        d_samp = (1 - u) * d_up + u * near.view(*shape2)
        pts = ray0 + d_samp[..., None] * ray_dir
        pick = torch.randperm(pts_tracks.shape[0])[:4096]
        sampled = torch.cat([pts_tracks[pick].view(1, -1, 3), pts.view(1, -1, 3)], dim=1)
    return d_pred, sdf_tracks[:, -1, 0], sampled, finish, trips
