"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/ls2fm.h
declares, level geometry equals the oracle's, the class surface / state_dict layout is the reference's,
and the product refuses CPU tensors instead of silently falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import ls2fm
from ls2fm import _lib, hashgrid
from ls2fm.options import make_options, Options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from ls2fm.models import base as lbase
from oracle import fields as OF
from conftest import ROOT, load_golden


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ls2fm.h")).read()
    declared = sorted(set(re.findall(r"\b(ls2fm_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no prototypes found in include/ls2fm.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ls2fm.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared          # the python binding covers the whole header
    abi = int(re.search(r"#define\s+LS2FM_ABI_VERSION\s+(\d+)", header).group(1))          # library, binding and header agree
    assert lib.ls2fm_abi_version() == _lib.ABI_VERSION == abi
    assert lib.ls2fm_status_string(-2) == b"unsupported configuration"
    assert b"starved" in lib.ls2fm_status_string(-5)
    assert _lib.async_error() == 0 and _lib.async_error(clear=True) == 0      # nothing asynchronous has run: no word, no error


def test_docs_carry_no_stale_abi_literal():
    """VERDICT r4 / r5 hygiene item: INTEGRATION.md's Level-3 stub twice shipped an `LS2FM_ABI_VERSION = <n>` literal that a
    later ABI bump left behind (the stub then fails its own assert).  Every such literal in the repo's documents must equal
    the header's value -- or, better, not be a literal at all."""
    header = open(os.path.join(ROOT, "include", "ls2fm.h")).read()
    abi = int(re.search(r"#define\s+LS2FM_ABI_VERSION\s+(\d+)", header).group(1))
    for doc in ("INTEGRATION.md", "README.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"LS2FM_ABI_VERSION\s*=\s*(\d+)", text):
            assert int(m.group(1)) == abi, f"{doc}: stale literal `{m.group(0)}` (include/ls2fm.h says {abi})"


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.GridDesc) == 4 * (2 + 16 * 4 + 17)
    assert ctypes.sizeof(_lib.FieldDesc) == 4 * (3 + 3 + 1 + 1 + 1 + 1 + 1 + 3 + 1 + 1)
    assert ctypes.sizeof(_lib.Linear) == 24
    assert ctypes.sizeof(_lib.Params) == 8 + 48 + 8 + 8 + 8 + 48 + 72 + 8
    assert ctypes.sizeof(_lib.ParamGrads) == 8 + 48 + 8 + 8 + 48 + 72
    assert ctypes.sizeof(_lib.LossSpec) == 12 * 8 and _lib.LossSpec.flags.offset == 88 and _lib.LossSpec.count_scale.offset == 92   # ls2fm_loss_spec: eleven pointers, uint32 flags, uint32 count_scale
    assert ctypes.sizeof(_lib.RenderOpts) == 8 + 8 + 8 + 4 * 8 + 8 + 8 + 8     # int32 (+pad), pointer, int32 (+pad), void* [4], void* x 3
    assert _lib.RenderOpts.loss_inputs_ready.offset == 56 and _lib.RenderOpts.depth_grad_ready.offset == 64
    assert _lib.RenderOpts.depth_bwd.offset == 72 and ctypes.sizeof(_lib.DepthBackward) == 48 and _lib.DepthBackward.workspace.offset == 40
    assert _lib.RenderOpts.loss.offset == 8 and _lib.RenderOpts.n_level_groups.offset == 16 and _lib.RenderOpts.group_events.offset == 24


@pytest.mark.parametrize("ds", ["DTU", "ETH3D", "BlendedMVS", "scannet"])
@pytest.mark.parametrize("L,log2_T", [(16, 19), (8, 10), (4, 11), (5, 14)])
def test_grid_desc_equals_oracle_table(ds, L, log2_T):
    cfg = OF.dataset_config(ds, n_levels=L, log2_hashmap_size=log2_T)
    t = cfg.table()
    d = hashgrid.build_grid_desc(L, 2, log2_T, 16, t.per_level_scale)
    assert np.array_equal(np.array(d.scale[:L], np.float32).view(np.uint32), t.scale.view(np.uint32))
    assert list(d.resolution[:L]) == list(t.resolution) and list(d.size[:L]) == list(t.size)
    assert list(d.offset[:L + 1]) == list(t.offset) and [bool(h) for h in d.hashed[:L]] == list(t.hashed)
    assert hashgrid.n_table_floats(d) == t.n_params


def test_state_dict_layout_is_the_references(manifest):
    for dual in (False, True):
        opt = make_options("DTU", device="cpu", dual_field=dual)
        ref = manifest["_state_dict_full_dtu"]["dual" if dual else "single"]
        assert {k: list(v.shape) for k, v in SDF(opt).state_dict().items()} == ref["sdf"]
        assert {k: list(v.shape) for k, v in RadF(opt).state_dict().items()} == ref["rad"]
    geo = manifest["_hash_geometry"]
    for ds in ("DTU", "ETH3D", "BlendedMVS", "scannet"):
        e = lbase.get_Embedder(make_options(ds, device="cpu"), input_dim=3, input_choice="Hash")
        assert e.out_dim == geo[ds]["out_dim"] == 35
        assert hashgrid.n_table_floats(e.embedder_obj.desc) == geo[ds]["n_params"]


def test_class_surface():
    opt = make_options("BlendedMVS", device="cpu", dual_field=True)
    sdf, rad, ren = SDF(opt), RadF(opt), Renderer(opt)
    for attr in ("bound_max", "bound_min", "center", "half_size", "rescale", "beta_speed", "beta", "sdf_threshold",
                 "iters_max", "scale_mlp", "embed_fn", "SDF_MLP"):
        assert hasattr(sdf, attr), attr
    for m in ("infer_sdf", "forward_ab", "sdf_to_sigma", "get_surface_pts", "gradient", "sphere_tracing"):
        assert callable(getattr(sdf, m))
    for attr in ("embed_fn", "Geo_enc", "embed_fn_v", "Rad_dec", "Geometry_feat", "infer_embed_v", "infer_app"):
        assert hasattr(rad, attr), attr
    for m in ("forward", "composite", "sample_depth", "volsdf_sampling", "sdf_to_sigma", "error_bound", "sample_pdf",
              "opacity_to_sample", "sample_depth_from_opacity"):
        assert callable(getattr(ren, m))
    assert tuple(sdf.bound_max.shape) == (1, 1, 3) and tuple(ren.center.shape) == (1, 1, 3)
    assert ren.bgcolor.tolist() == [1.0, 1.0, 1.0]
    assert sdf.iters_max == 20 and sdf.scale_mlp == 3.0
    a, b = sdf.forward_ab()
    assert abs(b.item() - 0.05) < 1e-7 and abs(a.item() - 20.0) < 1e-4
    # geometric init: hash columns of the first layer are zero, last-layer bias is -radius
    assert float(sdf.SDF_MLP.mlp[0].weight_v[:, 3:].abs().max()) == 0.0
    assert torch.allclose(sdf.SDF_MLP.mlp[1].bias, torch.full((17,), -1.0))
    assert rad.Rad_dec.mlp_radiance[0].weight_v.shape == (64, 65)
    opt1 = make_options("DTU", device="cpu", dual_field=False)
    assert RadF(opt1).Rad_dec.mlp_radiance[0].weight_v.shape == (64, 49)
    assert not hasattr(RadF(opt1), "Geo_enc")


def test_fourier_embedding_matches_reference_golden():
    g = load_golden("fourier")
    emb = lbase.get_Embedder(None, input_dim=3, input_choice="Fourier")
    out = emb(torch.from_numpy(g["d"]))
    assert out.shape[-1] == emb.out_dim == 27
    assert np.abs(out.numpy() - g["out"]).max() < 1e-6


def test_cpu_tensors_are_refused_not_silently_served():
    opt = make_options("DTU", device="cpu", hash_encoding=dict(n_levels=4, n_features_per_level=2,
                                                               log2_hashmap_size=10, base_resolution=16))
    sdf = SDF(opt)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sdf.infer_sdf(torch.zeros(4, 3))
    ren = Renderer(opt)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ren.forward(opt, torch.zeros(1, 2, 3), torch.ones(1, 2, 3), sdf, RadF(opt))


def test_composite_helper_matches_golden():
    """Renderer.composite is plain torch (used by the general form); check it against the reference"""
    g = load_golden("dtu_single")
    ren = Renderer(make_options("DTU", device="cpu"))
    rgb, prob = ren.composite(torch.from_numpy(g["comp_ray"]), torch.from_numpy(g["comp_rgb_s"]),
                              torch.from_numpy(g["comp_sig_s"]), torch.from_numpy(g["comp_t_s"]))
    assert np.abs(rgb.numpy() - g["comp_rgb"]).max() < 1e-5
    assert np.abs(prob.numpy() - g["comp_prob"]).max() < 1e-6


def test_options_attribute_dict():
    o = Options(a=dict(b=dict(c=3)), d=[1, 2])
    assert o.a.b.c == 3 and o["a"]["b"]["c"] == 3 and o.d == [1, 2]
    o.a.b.c = 4
    assert o.a.b.c == 4
    with pytest.raises(AttributeError):
        _ = o.missing


def test_loss_head_and_fused_adam_have_no_cpu_path():
    """the round's later additions keep the rule: CPU tensors are refused, nothing falls back"""
    import pytest
    import torch
    from ls2fm.losses import RenderLossHead
    from ls2fm.optim import FusedAdam
    ret = {"rgb": torch.zeros(1, 4, 3), "normals": torch.zeros(1, 4, 2, 3), "depth_mlp": torch.zeros(1, 4, 1)}
    head = RenderLossHead.__new__(RenderLossHead)
    head.weights = torch.ones(3)
    with pytest.raises(RuntimeError):
        head.terms(ret, torch.zeros(1, 4, 3))
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError):
        FusedAdam([p]).step()


def test_extract_mesh_lattice_is_the_references():
    """utils/util.py:399-409 by hand: idx -> (fmod(idx/N/N, N), fmod(idx/N, N), idx % N) under true division, times
    volume_size/(N-1), plus the origin in REVERSED column order"""
    from ls2fm.utils import util
    N, s = 8, 2.0
    pts = util.lattice_points(s, N, bound_max=[1.0, 2.0, 3.0], bound_min=[-1.0, -2.0, -3.0])
    assert pts.shape == (512, 3) and pts.dtype == np.float32
    for idx in (0, 1, 9, 70, 511):
        fx, fy, fz = ((idx / N) / N) % N, (idx / N) % N, idx % N
        want = np.array([fx * (s / 7) - 3.0, fy * (s / 7) - 2.0, fz * (s / 7) - 1.0]).astype(np.float32)
        assert np.array_equal(pts[idx], want), idx
    reg = util.lattice_points(s, N, reference_indexing=False)
    assert np.array_equal(reg[70], np.array([1 * s / 7 - 1, 0 * s / 7 - 1, 6 * s / 7 - 1]).astype(np.float32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        util.sdf_volume(SDF(make_options("DTU", device="cpu")), N=4)


def test_checkpoint_layout_round_trip(tmp_path, manifest):
    """model.ckpt in the reference's layout (utils/util.py:198-259): a checkpoint assembled key by key from the recorded
    state_dict manifest of the reference loads; one written here has exactly those keys and restores optimizer state"""
    from types import SimpleNamespace
    from ls2fm.utils import util
    opt = make_options("DTU", device="cpu", dual_field=True)
    opt.output_path = str(tmp_path)
    ref = manifest["_state_dict_full_dtu"]["dual"]
    gen = torch.Generator().manual_seed(0)
    ckpt = dict(epoch=None, iter=120,
                sdf_func={k: torch.randn(*shape, generator=gen) for k, shape in ref["sdf"].items()},
                color_func={k: torch.randn(*shape, generator=gen) for k, shape in ref["rad"].items()},
                cam_info={"se3": torch.zeros(2, 6)}, pts3d_info=None)
    torch.save(ckpt, tmp_path / "theirs.ckpt")
    model = SimpleNamespace(sdf_func=SDF(opt), color_func=RadF(opt))
    assert util.restore_checkpoint_sfm(opt, model, load_name=str(tmp_path / "theirs.ckpt")) == (None, None)
    for k, v in model.sdf_func.state_dict().items():
        assert torch.equal(v, ckpt["sdf_func"][k]), k
    for k, v in model.color_func.state_dict().items():
        assert torch.equal(v, ckpt["color_func"][k]), k
    assert torch.equal(model.cam_info_reloaded["se3"], torch.zeros(2, 6))
    # write -> read, with an optimizer and a scheduler riding along
    model.optim_sdf = torch.optim.Adam(model.sdf_func.parameters(), lr=1e-3)
    model.sched_sdf = torch.optim.lr_scheduler.ExponentialLR(model.optim_sdf, 0.9)
    for p in model.sdf_func.parameters():
        p.grad = torch.ones_like(p)
    model.optim_sdf.step(); model.sched_sdf.step()
    util.save_checkpoint_sfm(opt, model, ep=None, it=121, latest=False)
    assert os.path.exists(tmp_path / "model.ckpt") and os.path.exists(tmp_path / "model" / "121.ckpt")
    mine = torch.load(tmp_path / "model.ckpt", weights_only=False)
    assert {k: list(v.shape) for k, v in mine["sdf_func"].items()} == ref["sdf"]
    assert {k: list(v.shape) for k, v in mine["color_func"].items()} == ref["rad"]
    assert set(mine) == {"epoch", "iter", "sdf_func", "color_func", "cam_info", "pts3d_info", "optim_sdf", "sched_sdf"}
    other = SimpleNamespace(sdf_func=SDF(opt), color_func=RadF(opt))
    other.optim_sdf = torch.optim.Adam(other.sdf_func.parameters(), lr=1e-3)
    other.sched_sdf = torch.optim.lr_scheduler.ExponentialLR(other.optim_sdf, 0.9)
    assert util.restore_checkpoint_sfm(opt, other, resume=True) == (None, 121)
    assert util.restore_checkpoint_sfm(opt, other, resume=121) == (None, 121)
    for (k, a), b in zip(other.sdf_func.state_dict().items(), model.sdf_func.state_dict().values()):
        assert torch.equal(a, b), k
    assert other.optim_sdf.state_dict()["state"][0]["step"] == model.optim_sdf.state_dict()["state"][0]["step"]
    assert other.sched_sdf.last_epoch == 1


def test_render_size_limit_is_reported_not_overflowed():
    """more sample points than one call indexes with 32 bits: UNSUPPORTED from the size query (no launch, no overflow)"""
    lib = _lib.load()
    opt = make_options("DTU", device="cpu", dual_field=True, sample_intvs=128)
    from ls2fm import fused
    fdesc = fused.field_desc(opt)
    desc = SDF(opt).embed_fn.embedder_obj.desc
    ok = lib.ls2fm_render_workspace_bytes(ctypes.byref(fdesc), ctypes.byref(desc), 65536)          # 2^23 points: allowed
    assert ok > 0
    assert lib.ls2fm_render_workspace_bytes(ctypes.byref(fdesc), ctypes.byref(desc), 65537) == -2  # LS2FM_ERR_UNSUPPORTED


def test_mesh_sweep_lattices_are_the_references():
    """ls2fm.utils.plots: the axes / lattices of get_grid_uniform and get_grid (utils/plots.py:325-370) against lattices
    recorded from the reference's own functions (tests/golden/make_golden_plots.py); bit-exact (fp64 axes, fp32 points)"""
    from conftest import load_golden
    from ls2fm.utils import plots
    g = load_golden("plots_lattices")
    axes = plots.uniform_axes(7, [-0.6, 0.6])
    for a, ax in zip("xyz", axes):
        assert np.array_equal(ax, g[f"uniform/{a}"])
    assert np.array_equal(plots.lattice_on_device(axes, "cpu").numpy(), g["uniform/points"])
    for tag in ("fit_x", "fit_y", "fit_z"):
        axes, length, k = plots.fitted_axes(torch.from_numpy(g[f"{tag}/input"]), 9)
        assert k == int(g[f"{tag}/index"]) and length == float(g[f"{tag}/length"])
        for a, ax in zip("xyz", axes):
            assert np.array_equal(ax, g[f"{tag}/{a}"]), (tag, a)
        pts = plots.lattice_on_device(axes, "cpu")
        assert np.array_equal(pts.numpy(), g[f"{tag}/points"])
        part = plots.lattice_on_device(axes, "cpu", first=1234, count=777)           # chunks of the same lattice
        assert np.array_equal(part.numpy(), g[f"{tag}/points"][1234:1234 + 777])


@pytest.mark.parametrize("dataset", ["DTU", "ETH3D", "BlendedMVS", "scannet"])
def test_level_table_of_every_dataset_preset_matches_the_written_out_values(dataset):
    """The per-level (scale, resolution, size, offset, hashed) table of the shipped L16 / F2 / T2^19 grid for each of the four
    dataset presets against LITERAL values (tests/golden/level_tables.json, generated by make_level_tables.py with the
    reference's per-level scale).  The table is tcnn's, restated (parity unpinned): the file also names the levels whose
    resolution changes inside the error envelope of a device libm (log2f 1 ulp, exp2f 2 ulp) -- ETH3D 15 (10241 here, 10240
    possible), BlendedMVS 15, ScanNet 5 / 10 / 15 -- so a cross-check against a real tinycudann build knows where to look."""
    import json
    import os
    from conftest import GOLDEN
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    ref = json.load(open(os.path.join(GOLDEN, "level_tables.json")))[dataset]
    sdf = SDF(make_options(dataset, device="cpu"))
    enc = sdf.embed_fn.embedder_obj
    d = enc.desc
    assert d.n_levels == 16 and enc.params.numel() == ref["n_params"]
    sensitive = [lv["level"] for lv in ref["levels"] if lv["ulp_sensitive"]]
    for lv in ref["levels"]:
        l = lv["level"]
        got = (float(d.scale[l]), int(d.resolution[l]), int(d.size[l]), int(d.offset[l]), bool(d.hashed[l]))
        want = (lv["scale"], lv["resolution"], lv["size"], lv["offset"], lv["hashed"])
        assert got == want, (f"{dataset} level {l}: {got} != {want}; levels whose resolution a 1-2 ulp different log2f / exp2f changes "
                             f"(a real-tcnn table may differ THERE): {sensitive}, candidates {lv['resolutions_within_libm_envelope']}")
    assert sensitive == {"DTU": [], "ETH3D": [15], "BlendedMVS": [15], "scannet": [5, 10, 15]}[dataset]
    if dataset == "ETH3D":
        assert ref["levels"][15]["resolutions_within_libm_envelope"] == [10240, 10241] and int(d.resolution[15]) == 10241


def test_numa_binding_is_safe_without_a_gpu_topology(monkeypatch):
    """ls2fm.numa (round 6): the NUMA binding of a GPU process reads sysfs only and changes nothing when the topology is not there
    (this container) or the switch is off"""
    from ls2fm import numa
    assert numa._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    before = os.sched_getaffinity(0)
    monkeypatch.setenv("LS2FM_NUMA_BIND", "0")
    assert numa.bind_to_gpu_numa_node(0) is None and os.sched_getaffinity(0) == before
    monkeypatch.delenv("LS2FM_NUMA_BIND")
    node = numa.gpu_numa_node(0)
    if node is None:                                       # no KFD topology here: nothing is bound
        assert numa.bind_to_gpu_numa_node(0) is None and os.sched_getaffinity(0) == before
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")            # a UUID form: not resolved, not an error
    assert numa.gpu_numa_node(0) is None


def test_no_kernel_reads_through_the_dispatch_or_queue_pointer():
    """Round 6: the AQL queue lives in HOST memory; a kernel whose descriptor asks for the dispatch (or queue) pointer reads it from its
    waves over PCIe -- `shade_bwd` did (a private array promoted to LDS, indexed by the flat work-item id: ~290 uncached reads per
    launch, 123 us from the GPU's NUMA node, 135 us from the other socket).  No kernel of the shipped library may."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_host_reads", os.path.join(ROOT, "tools", "check_host_reads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, n = mod.offenders(_lib.LIB_PATH)
    assert n > 50, n                      # the parser found the library's kernels
    assert not bad, bad


def test_hand_issued_loads_stay_untouched_until_their_wait():
    """Advisor r5: `slab_accumulate_persistent_kernel` issues scalar loads and its unit claim by hand and completes them at a LATER
    s_waitcnt; the compiler does not know the destination registers are in flight.  `tools/check_async_regs.py` follows every path from
    each such instruction to its wait in the shipped library's ISA: nothing on the way may read or write the destination (a copy, a
    spill, a re-use would) -- run on every build, i.e. on every compiler bump."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    spec = importlib.util.spec_from_file_location("check_async_regs", os.path.join(ROOT, "tools", "check_async_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    errors, n_sload, n_atomic = mod.check(_lib.LIB_PATH)
    assert n_sload >= 16 and n_atomic == 4, (n_sload, n_atomic)       # four template instances, each with the four loads + the claim
    assert not errors, errors
    # the walker does flag a copy made before the wait (and follows a taken branch to find it)
    rows = [(0, "s_load_dword", "s8, s[2:3], 0x0"), (8, "s_cbranch_scc1", "1"), (12, "s_endpgm", ""),
            (16, "s_mov_b32", "s9, s8"), (20, "s_waitcnt", "lgkmcnt(0)"), (24, "s_add_u32", "s8, s8, 1")]
    at = {a: i for i, (a, _, _) in enumerate(rows)}
    bad = mod.walk(rows, at, 0, mod.regs("s8"), lambda a: "lgkmcnt(0)" in a)
    assert [b[0] for b in bad] == [16], bad


def test_fixed_point_conversion_through_the_doubles_mantissa_is_rint():
    """`slab_accumulate`'s add_fixed (csrc/bin_scatter.hip, LS2FM_ACC_CVT): fma((double) v, 2^shift, 1.5 * 2^52), reinterpreted, minus
    0x4338 << 48 -- must be the round-to-nearest-even integer of v * 2^shift (what __float2ll_rn gave) whenever |v * 2^shift| < 2^51.
    The device code is three instructions; this restates its arithmetic in numpy (the product is exact in double: fma = multiply, add)."""
    rng = np.random.default_rng(5)
    v = np.concatenate([
        rng.standard_normal(200000).astype(np.float32) * np.float32(3.0e-3),
        (rng.integers(-2**22, 2**22, 20000).astype(np.float32) + np.float32(0.5)) * np.float32(2.0**-30),      # exact ties
        np.array([0.0, -0.0, 1e-45, -1e-45, 2.0**-20, -(2.0**-20), np.nextafter(np.float32(2.0**-10), np.float32(0))], np.float32)])
    for shift in (30, 41, 20):
        scale = np.float64(2.0**shift)
        prod = v.astype(np.float64) * scale
        keep = np.abs(prod) < 2.0**51
        d = prod[keep] + np.float64(6755399441055744.0)
        bits = d.view(np.uint64) - np.uint64(0x4338000000000000)
        want = np.rint(prod[keep]).astype(np.int64)
        assert np.array_equal(bits.view(np.int64), want)
        assert keep.sum() > 200000
