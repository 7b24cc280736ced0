"""ls2fm_camera_rays (one launch) against the torch restatement of the reference's ray construction it replaces in the loops'
no-gradient preamble: `get_center_and_ray` (utils/camera.py:230-252), `Camera.get_pts3D`'s key-point rays
(pipelines/Camera.py:129-133) and `Lie.se3_to_SE3` (utils/camera.py:63-147) -- the host mirrors in ls2fm.utils.camera, which
tests/test_camera_helpers.py holds to the reference's formulas on the CPU."""
import math

import pytest
import torch

from ls2fm import stage
from ls2fm.utils import camera as cam

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(V=3, H=48, W=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    se3 = torch.cat([0.4 * torch.randn(V, 3, generator=g), 2.0 * torch.randn(V, 3, generator=g)], dim=1).to(DEV)
    se3[0, :3] = 0.0                                            # theta = 0: the series' first terms only
    intr = torch.tensor([[1.3 * W, 0.0, W / 2.0], [0.0, 1.2 * W, H / 2.0], [0.0, 0.0, 1.0]], device=DEV)
    return se3, intr, H, W, g


def _close(a, b, what):
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    same = float((a == b).float().mean())
    print(f"[camera_rays] {what}: max |diff| {err:.2e} of scale {scale:.2e}, bit-identical {100 * same:.1f} %")
    assert err <= 4e-7 * scale, what                           # a few ulp: the summation order of the 3- and 4-term products


def test_camera_rays_match_the_torch_ray_construction():
    se3, intr, H, W, g = _setup()
    kinv = cam.host_inverse_intrinsic(intr)
    poses = cam.lie.se3_to_SE3(se3)
    idx = torch.randperm(H * W, generator=g)[:500].to(DEV)
    grid = cam.mesh_grid(H=H, W=W, device=DEV)
    c_ref, r_ref = cam.get_center_and_ray(None, poses, intr=intr.unsqueeze(0), rays_idx=idx, xy_grid=grid)
    c, r = cam.camera_rays(kinv, poses=poses, pix=idx, width=W)
    _close(c, c_ref, "centers (poses, pixel indices)")
    _close(r, r_ref, "rays (poses, pixel indices)")
    # the exponential in the same launch
    out_poses = torch.zeros(se3.shape[0], 3, 4, device=DEV)
    c2, r2 = cam.camera_rays(kinv, se3=se3, pix=idx, width=W, poses_out=out_poses)
    _close(out_poses, poses, "se3 -> SE3")
    _close(c2, c_ref, "centers (se3)")
    _close(r2, r_ref, "rays (se3)")
    assert torch.equal(out_poses[0, :, :3], torch.eye(3, device=DEV))        # theta = 0: R = I exactly
    # key points of ONE view chosen by a device-side index, written into given buffers
    kp = (torch.rand(se3.shape[0], 37, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])).to(DEV)
    center, ray = torch.zeros(1, 37, 3, device=DEV), torch.zeros(1, 37, 3, device=DEV)
    for v in range(se3.shape[0]):
        sel = torch.tensor([v], device=DEV)
        cam.camera_rays(kinv, poses=poses, xy=kp, view_sel=sel, out=(center, ray))
        kc, kr = stage.keypoint_rays(poses[v], intr, kp[v])
        _close(center, kc, f"key-point centers, view {v}")
        _close(ray, kr, f"key-point rays, view {v}")


def test_fused_se3_exponential_and_its_gradient_match_autograd():
    se3, _, _, _, g = _setup(V=5, seed=3)
    se3 = se3.clone()
    se3[0, :3] = torch.tensor([1e-3, -2e-3, 5e-4])              # small angle: the series' low-order terms carry everything
    upstream = torch.randn(5, 3, 4, generator=g).to(DEV)
    a = se3.clone().requires_grad_(True)
    b = se3.clone().requires_grad_(True)
    pa, pb = cam.lie.se3_to_SE3(a), cam.se3_to_SE3_fused(b)
    _close(pb.detach(), pa.detach(), "se3 -> SE3 (autograd node)")
    (pa * upstream).sum().backward()
    (pb * upstream).sum().backward()
    scale = float(a.grad.abs().max())
    err = float((a.grad - b.grad).abs().max())
    print(f"[se3 exp] gradient: max |diff| {err:.2e} of scale {scale:.2e}")
    assert err <= 2e-6 * scale
    # against the closed forms in float64 (sin / cos): the 11-term series is exact to fp32 for these angles
    w = se3[:, :3].double().cpu()
    th = w.norm(dim=-1)[:, None, None]
    Wm = cam.skew(w)
    R64 = torch.eye(3, dtype=torch.float64) + torch.sin(th) / th * Wm + (1 - torch.cos(th)) / th ** 2 * (Wm @ Wm)
    assert float((pb.detach().cpu().double()[:, :, :3] - R64).abs().max()) < 5e-7


def test_camera_rays_refuses_cpu_tensors():
    se3, intr, H, W, g = _setup()
    kinv = cam.host_inverse_intrinsic(intr)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cam.camera_rays(kinv, poses=cam.lie.se3_to_SE3(se3).cpu(), pix=torch.arange(4), width=W)


def test_c_abi_argument_checks_of_the_round_4_entry_points():
    """LS2FM_ERR_INVALID_ARGUMENT (-1), nothing enqueued: both or neither of poses / se3, both or neither of xy / pix, a pixel
    index list without the image width, an unknown scatter mode"""
    import ctypes
    from ls2fm import _lib
    lib = _lib.load()
    se3, intr, H, W, g = _setup()
    kinv = cam.host_inverse_intrinsic(intr)
    poses = cam.lie.se3_to_SE3(se3).contiguous()
    idx = torch.arange(16, device=DEV)
    c, r = torch.empty(3, 16, 3, device=DEV), torch.empty(3, 16, 3, device=DEV)
    P = _lib.ptr
    call = lambda po, se, xy, pix, width: lib.ls2fm_camera_rays(P(po), P(se), kinv, P(xy), P(pix), width, 0, None, 3, 16, P(c), P(r), None,
                                                                _lib.stream_ptr())
    assert call(poses, se3, None, idx, W) == -1
    assert call(None, None, None, idx, W) == -1
    assert call(poses, None, None, None, W) == -1
    assert call(poses, None, None, idx, 0) == -1
    assert call(poses, None, None, idx, W) == 0
    assert lib.ls2fm_set_scatter_mode(7) == -1 and lib.ls2fm_get_scatter_mode() == 1
    assert lib.ls2fm_adam_sched_decay(1, None, _lib.stream_ptr()) == -1
    assert lib.ls2fm_se3_exp_fwd(None, 2, None, _lib.stream_ptr()) == -1
    assert lib.ls2fm_tracing_term_fwd(None, None, None, None, None, None, 4, None, _lib.stream_ptr()) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("use_sdfs", [False, True])
def test_fused_tracing_term_matches_the_torch_lines(use_sdfs):
    """ls2fm_tracing_term_fwd / _bwd against the torch lines they replace in TracingConsistency.__call__ (addcmul, live / sum,
    norm, dot, abs, dot): values and the gradients w.r.t. the traced depths and the last SDF values, incl. a key point that
    sits exactly on its target (norm backward at 0: zero) and padded (live = 0) rows"""
    from ls2fm.stage import _TracingTerm
    gen = torch.Generator().manual_seed(17)
    n = 1500
    center = torch.randn(1, n, 3, generator=gen).to(DEV)
    ray = torch.randn(1, n, 3, generator=gen).to(DEV)
    d0 = (torch.rand(1, n, generator=gen) * 3).to(DEV)
    target = torch.randn(n, 3, generator=gen).to(DEV)
    target[5] = (center[0, 5] + ray[0, 5] * d0[0, 5])
    live = (torch.rand(n, generator=gen) > 0.2).float().to(DEV)
    s0 = torch.randn(n, generator=gen).to(DEV)
    s0[7] = 0.0
    res = {}
    for which in ("fused", "torch"):
        d = d0.clone().requires_grad_(True)
        sl = s0.clone().requires_grad_(True)
        if which == "fused":
            terms = _TracingTerm.apply(center, ray, d, target, live, sl if use_sdfs else None)
            tl, ss = terms[0], terms[1]
        else:
            surface = torch.addcmul(center[0], ray[0], d.reshape(-1, 1))
            weight = live / live.sum()
            tl = torch.dot(torch.linalg.vector_norm(target - surface, dim=-1), weight)
            ss = torch.dot(sl.reshape(-1).abs(), weight)
        loss = 0.7 * tl + (0.3 * ss if use_sdfs else 0.0)
        loss.backward()
        res[which] = (tl.detach(), ss.detach(), d.grad.clone(), None if sl.grad is None else sl.grad.clone())
    f, t = res["fused"], res["torch"]
    assert abs(float(f[0]) - float(t[0])) < 2e-6 * abs(float(t[0]))
    assert torch.allclose(f[2], t[2], rtol=2e-5, atol=1e-9) and float(f[2][0, 5]) == 0.0
    if use_sdfs:
        assert abs(float(f[1]) - float(t[1])) < 2e-6 * abs(float(t[1]))
        assert torch.allclose(f[3], t[3], rtol=2e-5, atol=1e-9) and float(f[3][7]) == 0.0


@pytest.mark.parametrize("error,with_add", [(3.5, True), (12.0, True), (12.0, False), (10.0, False)])
def test_fused_ba_terms_match_the_torch_lines(error, with_add):
    """ls2fm_ba_terms_fwd / _bwd against BALoop._extra's torch lines (pipelines/BA.py:160-170): sdf_surf = mean |sdfs|, the adaptive
    weight from the DETACHED re-projection error (10^1 above 10 px, not AT 10), the weighted sum with the tracing loss -- values and
    the gradients w.r.t. the error, the SDF values (a zero among them: sign(0) = 0) and the added term"""
    from ls2fm import _lib
    from ls2fm.stage import _BATerms
    gen = torch.Generator().manual_seed(29)
    n = 1337
    s0 = torch.randn(n, 1, generator=gen).to(DEV) * 0.01
    s0[11] = 0.0
    w_surf, w_add = 10.0 ** 1.5, 10.0 ** 0.5
    res = {}
    for which in ("fused", "torch"):
        r = torch.tensor(error, device=DEV, requires_grad=True)
        sd = s0.clone().requires_grad_(True)
        a = torch.tensor(0.37, device=DEV, requires_grad=True) if with_add else None
        if which == "fused":
            surf, w, extra = _BATerms.apply(r, sd, a, 10.0, 1.0, 10.0, w_surf, w_add if with_add else 0.0)
            assert not surf.requires_grad and not w.requires_grad
        else:
            w = torch.where(r.detach() > 10, torch.full((), 10.0, device=DEV), torch.full((), 1.0, device=DEV))
            surf = sd.abs().mean()
            extra = w * r + w_surf * surf + (w_add * a if with_add else 0.0)
        (2.0 * extra).backward()
        res[which] = (surf.detach(), w.detach(), extra.detach(), r.grad.clone(), sd.grad.clone(), None if a is None else a.grad.clone())
    f, t = res["fused"], res["torch"]
    assert float(f[1]) == float(t[1]) == (10.0 if error > 10 else 1.0)
    assert abs(float(f[0]) - float(t[0])) < 2e-6 * abs(float(t[0])) and abs(float(f[2]) - float(t[2])) < 2e-6 * abs(float(t[2]))
    assert float(f[3]) == float(t[3])
    assert torch.allclose(f[4], t[4], rtol=2e-6, atol=0.0) and float(f[4][11]) == 0.0 and f[4].shape == s0.shape
    if with_add:
        assert abs(float(f[5]) - float(t[5])) < 1e-6 * abs(float(t[5]))
    lib = _lib.load()
    assert lib.ls2fm_ba_terms_fwd(None, None, 4, None, 10.0, 1.0, 10.0, 1.0, 0.0, None, None, None, _lib.stream_ptr()) == -1
    assert lib.ls2fm_ba_terms_bwd(None, 0, None, None, 1.0, 0.0, None, None, None, _lib.stream_ptr()) == -1


def test_fused_match_term_matches_the_torch_lines():
    """ls2fm_match_term_fwd / _bwd against the torch lines of InitLoop._extra they replace (Camera.py:136, 168-178 +
    Initialization.py:154-160): surface points, cross-view projection, pixel error and |sdf| means over both views, and the
    gradients w.r.t. the traced depths and the last SDF values"""
    from ls2fm.stage import _MatchTerm, _weighted_pair, host_intrinsic
    se3, intr, H, W, g = _setup()
    poses = cam.lie.se3_to_SE3(se3[:2]).contiguous()
    gen = torch.Generator().manual_seed(23)
    n = 700
    cen = [torch.randn(1, n, 3, generator=gen).to(DEV) * 0.2 for _ in range(2)]
    ray = [torch.nn.functional.normalize(torch.randn(1, n, 3, generator=gen), dim=-1).to(DEV) for _ in range(2)]
    kps = [(torch.rand(n, 2, generator=gen) * torch.tensor([W, H])).to(DEV) for _ in range(2)]
    d0 = [(torch.rand(1, n, generator=gen) * 2 + 1).to(DEV) for _ in range(2)]
    s0 = [torch.randn(n, generator=gen).to(DEV) * 0.01 for _ in range(2)]
    fixed = (torch.cat([c.reshape(-1, 3) for c in cen]).contiguous(), torch.cat([r.reshape(-1, 3) for r in ray]).contiguous(),
             torch.cat([kps[1], kps[0]]).contiguous(), torch.stack([poses[1], poses[0]]).contiguous(), host_intrinsic(intr), n)
    surf = torch.zeros(2, n, 3, device=DEV)
    res = {}
    for which in ("fused", "pair", "torch"):          # "pair": the loop's form -- the weighted sum as ONE node (ls2fm_weighted_pair_*),
        d = [t.clone().requires_grad_(True) for t in d0]      # whose two gradients reach the match node as the halves of one buffer
        sl = [t.clone().requires_grad_(True) for t in s0]
        if which != "torch":
            terms = _MatchTerm.apply(fixed, surf.view(-1, 3), *d, *sl)
            re, ss = terms[0], terms[1]
        else:
            errs, sdfs, pts_all = [], [], []
            for v in range(2):
                pts = cen[v] + ray[v] * d[v].reshape(1, -1, 1)
                o = 1 - v
                uv = cam.cam2img(cam.world2cam(pts, poses[o:o + 1]), intr.unsqueeze(0))
                uv = (uv / (uv[..., 2:] + 1e-6))[..., :2]
                errs.append((uv[0] - kps[o]).norm(dim=-1)); sdfs.append(sl[v].reshape(-1)); pts_all.append(pts[0].detach())
            re, ss = torch.cat(errs).mean(), torch.cat(sdfs).abs().mean()
            ref_surf = torch.stack(pts_all)
        total = _weighted_pair(re, ss, 0.3, 2.0) if which == "pair" else 0.3 * re + 2.0 * ss
        total.backward()
        res[which] = (re.detach(), ss.detach(), [t.grad.clone() for t in d], [t.grad.clone() for t in sl], total.detach())
    f, t = res["fused"], res["torch"]
    assert abs(float(f[0]) - float(t[0])) < 1e-5 * abs(float(t[0])) and abs(float(f[1]) - float(t[1])) < 1e-5 * abs(float(t[1]))
    assert torch.allclose(surf, ref_surf, rtol=1e-6, atol=1e-6)
    for a, b in zip(f[2] + f[3], t[2] + t[3]):
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-8), float((a - b).abs().max())
    pr = res["pair"]
    assert abs(float(pr[4]) - float(f[4])) <= 2e-7 * abs(float(f[4]))
    for a, b in zip(pr[2] + pr[3], f[2] + f[3]):      # same kernels, same upstream values: the same bits
        assert torch.equal(a, b)
