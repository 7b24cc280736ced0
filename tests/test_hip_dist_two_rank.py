"""Two processes on ONE GPU (gloo backend over device tensors: the box has a single MI355X, RCCL refuses two ranks per device):
the benchmark's data-parallel step -- rays sharded by view, fused render with the loss head inside (global counts),
backward, gradient all-reduce (one flat message; and the overlapped form: table slices reduced level group by level group
from inside the backward) -- must give every rank the gradients of the single-process run over all rays."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_TRANSIENT = ("Address already in use", "Connection refused", "Connection reset", "connect() timed out", "unhandled system error",
              "ncclSystemError", "ncclUnhandledCudaError", "store", "Socket Timeout")


def _spawn(fn, args, nprocs, port_at=1):
    """mp.spawn with ONE retry for failures of the rendezvous / communicator set-up (a spawned process group on a box that has just
    torn another one down: seen once in ~80 spawns of a full-suite run, never in isolation) -- on a fresh port.  An assertion of the
    worker (a NUMERIC disagreement) is never retried.  Every first failure is written to gpurun_out/flaky_dist.txt."""
    try:
        mp.spawn(fn, args=args, nprocs=nprocs, join=True)
        return
    except Exception as e:                                        # noqa: BLE001
        text = f"{type(e).__name__}: {e}"
        try:
            out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "flaky_dist.txt"), "a") as f:
                f.write(f"==== {fn.__name__} {args}\n{text}\n")
        except OSError:
            pass
        if "AssertionError" in text or not any(t in text for t in _TRANSIENT):
            raise
    args = list(args)
    args[port_at] = _free_port()
    mp.spawn(fn, args=tuple(args), nprocs=nprocs, join=True)


def _step(opt, sdf, rad, ren, head, center, ray, gt, dref, masks):
    for p in list(sdf.parameters()) + list(rad.parameters()):
        p.grad = None
    ret, loss = ren.forward_with_loss(opt, center, ray, sdf, rad, head, gt, d_points=dref, **masks)
    loss["all"].backward()
    return loss


def _worker(rank, world, port, overlap, out_dir, backend="gloo"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "level-s2fm_official_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      LS2FM_DIST_SINGLE="1" if world == 1 else "0")
    torch.cuda.set_device(0)
    dev = "cuda:0"
    from ls2fm import dist as ldist
    from ls2fm.losses import RenderLossHead
    from ls2fm.options import make_options
    from test_hip_fused_render import _randomized, _rays
    from helpers import named_grads
    n_rays = 256
    opt = make_options("BlendedMVS", device=dev, dual_field=True, sample_intvs=32)
    sdf, rad, ren = _randomized(opt, 111)                                   # identical replicas on every rank
    center, ray = _rays(n_rays, 2.0, 112)
    g = torch.Generator().manual_seed(113)
    gt = torch.rand(1, n_rays, 3, generator=g).to(dev)
    dref = (torch.rand(1, n_rays, generator=g) * 4).to(dev)
    mfin = (torch.rand(1, n_rays, generator=g) < 0.5).to(dev)
    mfin[:, : n_rays // 2] = False                                          # unbalanced between the ranks
    mbg = (torch.rand(1, n_rays, generator=g) < 0.8).to(dev)
    head = RenderLossHead(dev, 3.0, 2.0, 1.0, global_counts="allreduce")
    # ---- single-process reference over all rays (before the process group exists)
    full_loss = _step(opt, sdf, rad, ren, head, center, ray, gt, dref, dict(mask_finish=mfin, mask_eik=mbg, mask_bg=mbg))
    full = {**{"s." + k: v for k, v in named_grads(sdf).items()}, **{"r." + k: v for k, v in named_grads(rad).items()}}
    full_terms = {k: float(v) for k, v in full_loss.items()}
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        if overlap:
            ldist.enable_table_overlap(sdf, rad, n_groups=3)
        sl = slice(rank * n_rays // world, (rank + 1) * n_rays // world)
        c, r = ldist.shard_rays(center, ray)
        assert torch.equal(c, center[:, sl])
        loss = _step(opt, sdf, rad, ren, head, c.contiguous(), r.contiguous(), gt[:, sl].contiguous(), dref[:, sl].contiguous(),
                     dict(mask_finish=mfin[:, sl], mask_eik=mbg[:, sl], mask_bg=mbg[:, sl]))
        params = list(sdf.parameters()) + list(rad.parameters())
        red = ldist.GradAllReducer(params)
        assert red._all_in_flat() is not None                     # one flat buffer: no packing
        assert (getattr(red._all_in_flat(), "_ls2fm_pending", None) is not None) == overlap
        red.all_reduce()
        torch.cuda.synchronize()
        if backend == "nccl" and overlap:
            assert ldist._COALESCED["ok"] is True                 # RCCL: a level group's two slices went out as one launch
        got = {**{"s." + k: v for k, v in named_grads(sdf).items()}, **{"r." + k: v for k, v in named_grads(rad).items()}}
        from conftest import rel_err
        for k in full:
            assert rel_err(got[k], full[k]) < (2e-4 if k == "s.beta" else 2e-5), k
        for k in ("rgb_loss", "eikonal_loss", "DC_loss", "all"):          # every rank reports the global means
            assert abs(float(loss[k]) - full_terms[k]) <= 1e-5 * max(1.0, abs(full_terms[k])), k
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_one_gpu_reproduce_the_single_process_gradients(overlap, tmp_path):
    _spawn(_worker, (2, _free_port(), overlap, str(tmp_path)), 2)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _capture_overlap_worker(rank, world, port, out_dir):
    """the level-group reductions recorded INSIDE a hipGraph capture (ls2fm.dist.enable_capture_overlap): the
    captured step forks capturing stream -> communication stream -> RCCL's stream for the first group and joins at all_reduce()"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "level-s2fm_official_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", LS2FM_DIST_SINGLE="1")
    torch.cuda.set_device(0)
    dev = "cuda:0"
    from ls2fm import dist as ldist
    from ls2fm.graph import CapturedStep
    from ls2fm.losses import RenderLossHead
    from ls2fm.options import make_options
    from test_hip_fused_render import _randomized, _rays
    from helpers import named_grads
    n_rays = 512
    opt = make_options("ETH3D", device=dev, dual_field=True, sample_intvs=128)       # (>= 32 k samples: side jobs inside the fill)
    sdf, rad, ren = _randomized(opt, 121)
    center, ray = _rays(n_rays, float(opt.data.bound_max[0]), 122)
    gt = torch.rand(1, n_rays, 3, generator=torch.Generator().manual_seed(123)).to(dev)
    head = RenderLossHead(dev, 3.0, 2.0, 0.0, global_counts="uniform")
    params = list(sdf.parameters()) + list(rad.parameters())
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    try:
        ldist.enable_table_overlap(sdf, rad, n_groups=2)
        red = ldist.GradAllReducer(params)
        s_main = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(s_main)

        def whole_step():
            _step(opt, sdf, rad, ren, head, center, ray, gt, None, {})
            red.all_reduce()
        whole_step()
        torch.cuda.synchronize()
        want = {**{"s." + k: v.clone() for k, v in named_grads(sdf).items()}, **{"r." + k: v.clone() for k, v in named_grads(rad).items()}}
        ldist.enable_capture_overlap(True)
        cap = CapturedStep(whole_step, params, stream=s_main)
        for _ in range(3):
            for p in params:
                if p.grad is not None:
                    p.grad.zero_()
            cap.replay()
        torch.cuda.synchronize()
        got = {**{"s." + k: v for k, v in named_grads(sdf).items()}, **{"r." + k: v for k, v in named_grads(rad).items()}}
        for k in want:
            assert torch.equal(got[k], want[k]), k          # same kernels, same level groups: the same bits
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_level_group_reductions_inside_a_capture(tmp_path):
    _spawn(_capture_overlap_worker, (1, _free_port(), str(tmp_path)), 1)
    assert (tmp_path / "ok0").exists()


def test_one_rank_rccl_overlapped_reduction(tmp_path):
    """the same step through a real RCCL communicator (world size 1: all this box allows): the overlapped reduction's streams,
    events, coalesced launches and async handles run against the backend the multi-GPU bench uses; the sums are identities"""
    _spawn(_worker, (1, _free_port(), True, str(tmp_path), "nccl"), 1)
    assert (tmp_path / "ok0").exists()


def _stage_worker(rank, world, port, out_dir, backend, async_gather, shard_groups=1):
    """8 steps of the sharded stage (reduce-scatter -> Adam on this rank's shard -> all-gather) against 8 steps of the
    single-process RenderStage over all rays: same parameter trajectory"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "level-s2fm_official_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      LS2FM_DIST_SINGLE="1" if world == 1 else "0")
    torch.cuda.set_device(0)
    dev = "cuda:0"
    from ls2fm import stage, dist as ldist
    from ls2fm.options import make_options
    from test_hip_fused_render import _randomized, _rays
    n_rays, steps = 128, 8
    opt = make_options("DTU", device=dev, dual_field=True, sample_intvs=32,
                       hash_encoding=dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16))
    w = dict(rgb=3, eikonal_loss=1, DC_Loss=0)
    batches = []
    for it in range(steps):
        c, r = _rays(n_rays, 1.0, 300 + it)
        g = torch.Generator().manual_seed(400 + it)
        batches.append((c.view(2, n_rays // 2, 3), r.view(2, n_rays // 2, 3), torch.rand(2, n_rays // 2, 3, generator=g).to(dev)))
    # ---- single process, all rays
    sdf_a, rad_a, ren = _randomized(opt, 211)
    st_a = stage.RenderStage(opt, ren, sdf_a, rad_a, weights=w, lr=2e-3, lr_end=2e-4, max_iter=steps, lr_color=1e-3, eps=1e-15)
    ref_losses = [float(st_a.step(*b)["loss_all"]) for b in batches]
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        sdf_b, rad_b, _ = _randomized(opt, 211)
        st_b = stage.RenderStage(opt, ren, sdf_b, rad_b, weights=w, lr=2e-3, lr_end=2e-4, max_iter=steps, lr_color=1e-3, eps=1e-15,
                                 sharded=True, async_gather=async_gather, shard_groups=shard_groups)
        assert st_b.optim.world == world and st_b.optim.shard * world == st_b.optim.total
        assert st_b.optim.n_groups == shard_groups
        losses = []
        for c, r, gt in batches:
            cs, rs = ldist.shard_rays(c, r)                     # by view: one view per rank
            gs = gt[rank * (2 // world):(rank + 1) * (2 // world)] if world > 1 else gt
            losses.append(float(st_b.step(cs.contiguous(), rs.contiguous(), gs.contiguous())["loss_all"]))
        st_b.optim.wait_params()
        torch.cuda.synchronize()
        if shard_groups == 1:
            n_state = sum(s_["exp_avg"].numel() for s_ in st_b.optim.inner.state.values())
            assert n_state <= st_b.optim.shard                      # optimizer state / world
        else:                                                   # pipelined: table slices / world + the replicated small tensors
            pp = st_b.optim._pipe
            n_state = sum(s_["exp_avg"].numel() for o in pp["inner"] if o is not None for s_ in o.state.values())
            assert n_state == sum(pc["shard"] for grp in pp["pieces"] for pc in grp) <= st_b.optim.shard
        from conftest import rel_err
        for (k, pa), (_, pb) in zip(list(sdf_a.named_parameters()) + list(rad_a.named_parameters()),
                                    list(sdf_b.named_parameters()) + list(rad_b.named_parameters())):
            tol = 5e-2 if not k.endswith("embedder_obj.params") else 2e-1          # the bars of test_stage_trajectory_matches_plain_torch
            assert rel_err(pb, pa) < tol, (k, rel_err(pb, pa))
        for a, b in zip(losses, ref_losses):                    # every rank reports the global loss
            assert abs(a - b) <= 3e-3 * abs(b), (losses, ref_losses)
        with open(os.path.join(out_dir, f"stage_ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("async_gather", [False, True])
def test_two_ranks_sharded_stage_matches_single_process_trajectory(async_gather, tmp_path):
    _spawn(_stage_worker, (2, _free_port(), str(tmp_path), "gloo", async_gather), 2)
    assert (tmp_path / "stage_ok0").exists() and (tmp_path / "stage_ok1").exists()


@pytest.mark.parametrize("async_gather", [False, True])
def test_two_ranks_pipelined_sharded_stage_matches_single_process_trajectory(async_gather, tmp_path):
    """ShardedAdam(n_groups = 2) under RenderStage: a traced-depth node rides in the render's backward, so the per-group chain
    (reduce-scatter -> Adam on the slice -> all-gather, small tensors replicated) is issued at step(); same 8-step trajectory"""
    _spawn(_stage_worker, (2, _free_port(), str(tmp_path), "gloo", async_gather, 2), 2)
    assert (tmp_path / "stage_ok0").exists() and (tmp_path / "stage_ok1").exists()


def _pipelined_worker(rank, world, port, out_dir, backend, n_groups):
    """the benchmark's step -- fused render with the loss head inside, backward, update -- with the exchange issued from INSIDE the
    backward, level group by level group (ShardedAdam's hook on the table Parameter): 6 steps against the single-process run
    over all rays with FusedAdam"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "level-s2fm_official_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      LS2FM_DIST_SINGLE="1" if world == 1 else "0")
    torch.cuda.set_device(0)
    dev = "cuda:0"
    from ls2fm import dist as ldist
    from ls2fm.losses import RenderLossHead
    from ls2fm.optim import FusedAdam
    from ls2fm.options import make_options
    from test_hip_fused_render import _randomized, _rays
    n_rays, steps = 128, 6
    opt = make_options("BlendedMVS", device=dev, dual_field=True, sample_intvs=32,
                       hash_encoding=dict(n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=16))
    data = []
    for it in range(steps):
        c, r = _rays(n_rays, 2.0, 500 + it)
        gt = torch.rand(1, n_rays, 3, generator=torch.Generator().manual_seed(600 + it)).to(dev)
        data.append((c, r, gt))
    head = RenderLossHead(dev, 3.0, 2.0, None, global_counts="allreduce")

    def one(sdf, rad, ren, params, c, r, gt):
        for p in params:
            p.grad = None
        ret, loss = ren.forward_with_loss(opt, c, r, sdf, rad, head, gt)
        loss["all"].backward()
        return float(loss["all"])

    sdf_a, rad_a, ren = _randomized(opt, 77)
    pa = list(sdf_a.parameters()) + list(rad_a.parameters())
    ref_opt = FusedAdam([dict(params=list(sdf_a.parameters()), lr=2e-3), dict(params=list(rad_a.parameters()), lr=1e-3)], eps=1e-15,
                        scheduled_gamma=0.9)
    ref_losses = []
    for c, r, gt in data:
        ref_losses.append(one(sdf_a, rad_a, ren, pa, c, r, gt))
        ref_opt.step()
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        sdf_b, rad_b, _ = _randomized(opt, 77)
        so = ldist.ShardedAdam.for_fields(sdf_b, rad_b, lr=2e-3, lr_color=1e-3, eps=1e-15, scheduled_gamma=0.9, n_groups=n_groups,
                                          async_gather=True)
        assert so.n_groups == n_groups and so._pipe is not None
        pb = list(so.params)
        losses = []
        sl = slice(rank * n_rays // world, (rank + 1) * n_rays // world)
        for c, r, gt in data:
            so.wait_params()
            losses.append(one(sdf_b, rad_b, ren, pb, c[:, sl].contiguous(), r[:, sl].contiguous(), gt[:, sl].contiguous()))
            assert so._pipe["launched"], "the fused backward did not hand its level groups to the optimizer"
            so.step()
        so.wait_params()
        torch.cuda.synchronize()
        from conftest import rel_err
        for (k, p_a), (_, p_b) in zip(list(sdf_a.named_parameters()) + list(rad_a.named_parameters()),
                                      list(sdf_b.named_parameters()) + list(rad_b.named_parameters())):
            tol = 5e-2 if not k.endswith("embedder_obj.params") else 2e-1
            assert rel_err(p_b, p_a) < tol, (k, rel_err(p_b, p_a))
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= 3e-3 * abs(b), (losses, ref_losses)
        # a second gradient producer in such a step is refused (the exchanged buffer is no longer what autograd holds)
        so.wait_params()
        c, r, gt = data[0]
        one(sdf_b, rad_b, ren, pb, c[:, sl].contiguous(), r[:, sl].contiguous(), gt[:, sl].contiguous())
        pb[0].grad = pb[0].grad + 1.0
        try:
            so.step()
            refused = False
        except RuntimeError as e:
            refused = "only gradient producer" in str(e)
        so.wait_params()
        assert refused
        with open(os.path.join(out_dir, f"pipe_ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_groups", [2, 3])
def test_two_ranks_pipelined_exchange_from_inside_the_backward(n_groups, tmp_path):
    _spawn(_pipelined_worker, (2, _free_port(), str(tmp_path), "gloo", n_groups), 2)
    assert (tmp_path / "pipe_ok0").exists() and (tmp_path / "pipe_ok1").exists()


def test_one_rank_rccl_pipelined_exchange(tmp_path):
    """the pipelined form against a real RCCL communicator (world size 1): reduce_scatter_tensor / all_gather_into_tensor per
    level group on the communication stream, events recorded inside ls2fm_render_bwd, Adam on the communication stream"""
    _spawn(_pipelined_worker, (1, _free_port(), str(tmp_path), "nccl", 2), 1)
    assert (tmp_path / "pipe_ok0").exists()


def test_one_rank_rccl_sharded_stage(tmp_path):
    """the sharded step's collectives (reduce_scatter_tensor, all_gather_into_tensor in place, the communication stream) against
    a real RCCL communicator (world size 1: all this box allows)"""
    _spawn(_stage_worker, (1, _free_port(), str(tmp_path), "nccl", True), 1)
    assert (tmp_path / "stage_ok0").exists()


# ---- the stage LOOPS under data parallelism (BASELINE.json configs[3]: a Neural BA step with the rays of the registered views
# sharded over the GPUs, RCCL gradient all-reduce; pipelines/BA.py:110-188)
def _loop_worker(rank, world, port, out_dir, backend, which, capture):
    """5 iterations of BALoop / RefineLoop with the render rays sharded by view over `world` ranks (every rank: the same views,
    weights and per-iteration picks; point side / re-projection / tracing consistency replicated with weight 1 / world in the
    backward; ONE gradient all-reduce per iteration: the fields' flat buffer + the pose groups) against the single-process loop"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "level-s2fm_official_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      LS2FM_DIST_SINGLE="1" if world == 1 else "0")
    torch.cuda.set_device(0)
    import numpy as np
    from conftest import load_golden
    from ls2fm import stage
    from test_hip_stage_loops import _scene
    n_it = 5

    def build(distributed, cap):
        g = load_golden("stage_ba_dtu_dual" if which == "ba" else "stage_refine_dtu_dual")
        meta, opt, sdf, rad, ren, views, picks = _scene(g)
        o = meta["optim"]
        if which == "ba":
            views.poses = torch.from_numpy(g["se3"]).to("cuda")
            loop = stage.BALoop(opt, ren, sdf, rad, views, weights=meta["weights"], lr_sdf=o["lr_sdf"], lr_sdf_end=o["lr_sdf_end"],
                                lr_color=o["lr_color"], lr_pose_r=o["lr_pose_r"], lr_pose_t=o["lr_pose_t"], max_iter=o["max_iter"],
                                rand_rays=meta["rand_rays"], capture=cap, distributed=distributed)
        else:
            loop = stage.RefineLoop(opt, ren, sdf, rad, views, weights=meta["weights"], lr_sdf=o["lr_sdf"], lr_sdf_end=o["lr_sdf_end"],
                                    lr_color=o["lr_color"], max_iter=o["max_iter"], rand_rays=meta["rand_rays"], capture=cap,
                                    distributed=distributed)
        return loop, sdf, rad, picks

    loop_a, sdf_a, rad_a, picks = build(False, capture)
    logs_a = {k: v.cpu().numpy() for k, v in loop_a.run(n_iters=n_it, picks=picks).items()}
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        loop_b, sdf_b, rad_b, _ = build(True, capture)
        assert loop_b.stage.reducer is not None and loop_b.stage.shard_views
        logs_b = {k: v.cpu().numpy() for k, v in loop_b.run(n_iters=n_it, picks=picks).items()}
        torch.cuda.synchronize()
        assert (loop_b.stage._graph is not None) == capture
        for k in ("all", "PSNR", "rgb_loss", "eikonal_loss"):               # every rank reports the GLOBAL value
            assert np.all(np.abs(logs_b[k] - logs_a[k]) <= 3e-3 * np.abs(logs_a[k]) + 1e-6), (k, logs_b[k], logs_a[k])
        for k in ("sdf_surf", "tracing_loss") + (("reproj_error",) if which == "ba" else ()):
            assert np.all(np.abs(logs_b[k] - logs_a[k]) <= 2e-2 * np.abs(logs_a[k]) + 2e-4), (k, logs_b[k], logs_a[k])
        from conftest import rel_err
        for (k, pa), (_, pb) in zip(list(sdf_a.named_parameters()) + list(rad_a.named_parameters()),
                                    list(sdf_b.named_parameters()) + list(rad_b.named_parameters())):
            if not k.endswith("embedder_obj.params"):
                assert rel_err(pb, pa) < 5e-2, (k, rel_err(pb, pa))
        if which == "ba":
            a, b = loop_a.poses_se3().cpu(), loop_b.poses_se3().cpu()
            assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()), "poses"
        with open(os.path.join(out_dir, f"loop_ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["ba", "refine"])
def test_two_ranks_view_sharded_loops_match_the_single_process_trajectory(which, tmp_path):
    _spawn(_loop_worker, (2, _free_port(), str(tmp_path), "gloo", which, False), 2)
    assert (tmp_path / "loop_ok0").exists() and (tmp_path / "loop_ok1").exists()


@pytest.mark.parametrize("which", ["ba", "refine"])
def test_one_rank_rccl_captured_loops(which, tmp_path):
    """the CAPTURED iteration under a real RCCL communicator (world size 1: what this box allows): the loss-count, trip-count and
    gradient all-reduces are recorded into the iteration's hipGraph (tracings on the capturing stream: no side-stream branch forks
    again into RCCL's stream) -- same trajectory as the captured single-process loop"""
    _spawn(_loop_worker, (1, _free_port(), str(tmp_path), "nccl", which, True), 1)
    assert (tmp_path / "loop_ok0").exists()
