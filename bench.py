#!/usr/bin/env python
"""Benchmark of the Level-S2fM render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic rays, forward + backward (launched eagerly or as a
hipGraph replay, see --launch):
Renderer.forward (fused HIP kernels) -> loss head -> loss.backward() (fused HIP backward), and for N > 1 the
RCCL all-reduce of the gradients.  Workload (BASELINE.json configs[1]): ETH3D bounds (+-5, scale_mlp 5,
inside = false), 1024 rays x 128 samples per GPU, full L16/F2/T19 hash grids, dual field (SDF + radiance grid),
synthetic inputs per BASELINE.md section 3.  Weak scaling: every rank renders its own 1024 rays.

Prints ONE JSON line on rank 0 (contract in the task prompt) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "level-s2fm_official_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
F32_PEAK_TFLOPS = 157.3        # f32 vector / f32 MFMA peak (same number on gfx950)


def synthetic_rays(n_rays, s, device, seed=0):
    """center = (0,0,-2.5 s); ray = (0,0,1) + 0.15 N(0,1), unnormalised"""
    g = torch.Generator().manual_seed(seed)
    center = torch.tensor([0.0, 0.0, -2.5 * s]).repeat(1, n_rays, 1)
    ray = torch.tensor([0.0, 0.0, 1.0]).repeat(1, n_rays, 1) + 0.15 * torch.randn(1, n_rays, 3, generator=g)
    return center.to(device), ray.to(device)


def randomize(modules, seed=0):
    """tables U(-0.1, 0.1), first-layer hash columns N(0, 0.05): the hash path is live (SURVEY C-12)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in modules:
            for name, p in mod.named_parameters():
                if name.endswith("embedder_obj.params"):
                    p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * 0.1).to(p.device))
                if name.endswith("mlp.0.weight_v") and "Rad_dec" not in name:
                    p[:, 3:] = (torch.randn(p[:, 3:].shape, generator=g) * 0.05).to(p.device)
                    gname = name.replace("weight_v", "weight_g")
                    dict(mod.named_parameters())[gname].copy_(p.norm(dim=1, keepdim=True))


def loss_head(ret):
    """10^3 L1(rgb, 0.5) + 10^2 L1(|n|, 1) + smoothL1(depth)   (weights: options/LevelS2fM.yaml:102-107)
    plain-torch form: used by the CPU baseline leg and by --torch-loss"""
    rgb = (ret["rgb"] - 0.5).abs().mean()
    eik = (ret["normals"].norm(dim=-1) - 1.0).abs().mean()
    dep = torch.nn.functional.smooth_l1_loss(ret["depth_mlp"], torch.zeros_like(ret["depth_mlp"]))
    return 1e3 * rgb + 1e2 * eik + dep


def host_threads(cap=32):
    """cores this process may actually run on (affinity mask and cgroup quota, not the machine's core count: sizing the
    OpenMP pool by os.cpu_count() inside a CPU-limited container oversubscribes spinning threads)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline(dataset, dual, n_samples, n_rays=1024, budget_s=25.0):
    """The CPU oracle (a torch restatement of the reference's op sequence, pinned against the reference's golden vectors)
    timed on the host cores of this box on a bounded sample of the SAME workload: whole fwd+bwd steps of the benchmark's
    own batch (1024 rays x 128 samples, full-size grids) on all granted cores, plus one single-thread step of a quarter
    batch (SURVEY 8d asks for both figures).  The step cost has a large batch-independent part (gradients of the two
    12 M-entry tables), so small samples understate the CPU: 128 rays run at ~35 rays/s, 1024 rays at ~110."""
    from oracle import fields as OF
    threads = host_threads()
    cfg = OF.dataset_config(dataset, dual_field=dual, sample_intvs=n_samples)
    gen = torch.Generator().manual_seed(0)
    sd, rd = OF.init_sdf_state(cfg, gen), OF.init_rad_state(cfg, gen)
    OF.randomize_state(sd, gen)
    OF.randomize_state(rd, gen)
    sd = OF.to_dtype(sd, torch.float32, True)
    rd = OF.to_dtype(rd, torch.float32, True)
    s = cfg.bound_max[0]
    table = cfg.table()

    def step(rays):
        center, ray = synthetic_rays(rays, s, "cpu")
        for st in (sd, rd):
            for v in st.values():
                v.grad = None
        t = time.perf_counter()
        loss_head(OF.render(cfg, center, ray, sd, rd, table, table)).backward()
        return time.perf_counter() - t

    torch.set_num_threads(threads)
    step(64)                                  # warm-up (allocators, thread pool)
    times = []
    while len(times) < 1 or (sum(times) + times[-1] < budget_s and len(times) < 10):
        times.append(step(n_rays))
    dt = sorted(times)[len(times) // 2]
    torch.set_num_threads(1)
    n1 = max(32, n_rays // 4)
    dt1 = step(n1)
    torch.set_num_threads(threads)
    return {"value": n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} fwd+bwd step(s) of {n_rays} rays x {n_samples} samples (the benchmark's own batch and field "
                      f"config, torch CPU, {threads} threads), median {dt * 1e3:.0f} ms/step",
            "single_thread": {"value": n1 / dt1, "unit": "rays/s", "cores": 1,
                              "sample": f"1 fwd+bwd step of {n1} rays x {n_samples} samples, 1 thread, {dt1 * 1e3:.0f} ms"}}


# per-GPU workloads of BASELINE.json's configurations; weak scaling: every rank renders `rays` rays
CONFIGS = {
    "C1": dict(dataset="DTU", rays=256, samples=32, dual=False,
               note="configs[0]: one synthetic view, 256 rays x 32 samples, single field (the reference's CPU-runnable case)"),
    "C2": dict(dataset="ETH3D", rays=1024, samples=128, dual=True,
               note="configs[1]: ETH3D two-view init, 1024 rays x 128 samples, hash-grid SDF + radiance"),
    "C3": dict(dataset="DTU", rays=8192, samples=128, dual=True,
               note="configs[2]: DTU, dual field, the pipeline's 8192 rays per step (split over the views of a stage)"),
    "C4": dict(dataset="BlendedMVS", rays=1024, samples=128, dual=False,
               note="configs[3]: BlendedMVS Neural-BA step, one registered view (1024 rays) per GPU"),
    "C5": dict(dataset="scannet", rays=4096, samples=256, dual=False,
               note="configs[4]: ScanNet stress, 4096 rays x 256 samples with eikonal loss"),
}


def mirror_upkeep_us(sdf, rad, dev, reps=20):
    """extra device time of one FusedAdam step for keeping the interleaved table copy current (paired job vs two plain jobs)"""
    from ls2fm.optim import FusedAdam
    tabs = [sdf.embed_fn.embedder_obj.params, rad.embed_fn.embedder_obj.params]
    if any(getattr(t, "_ls2fm_mirror", None) is None for t in tabs):
        return None
    for t in tabs:
        if t.grad is None:
            t.grad = torch.zeros_like(t)

    def timed(opt):
        for _ in range(3):
            opt.step()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            opt.step()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    with_copy = timed(FusedAdam(tabs, lr=0.0))
    held = [t._ls2fm_mirror for t in tabs]
    for t in tabs:
        t._ls2fm_mirror = None
    try:
        plain = timed(FusedAdam(tabs, lr=0.0))
    finally:
        for t, h in zip(tabs, held):
            t._ls2fm_mirror = h
    return with_copy - plain


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2",
                    help="BASELINE.json workload (SURVEY 8d): C2 = the metric's configuration (default); C3-C5 per-GPU shapes of "
                         "configs[2..4]; --rays / --samples / --dataset / --single-field override single values")
    ap.add_argument("--rays", type=int, default=None, help="rays per GPU")
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--single-field", action="store_true", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # child process of the default run
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group, gradient all-reduce, barriers) even with one process")
    ap.add_argument("--launch", choices=("auto", "eager", "graph"), default="auto",
                    help="how the timed steps are launched: eager kernel launches, hipGraph replays of the whole step "
                         "(ls2fm.graph.CapturedStep), or whichever a short untimed probe finds faster on this host (eager "
                         "wins by ~5 %% when the host keeps up, the graph wins when the host CPU is slow or busy)")
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--torch-loss", action="store_true", help="loss head as separate PyTorch ops instead of the fused kernel")
    ap.add_argument("--overlap-groups", type=int, default=2, help="N > 1: level groups of the overlapped table-gradient reduction (2..4)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one gradient all-reduce after the backward instead of the "
                    "level-group reductions issued from inside it")
    ap.add_argument("--capture-overlap", action="store_true",
                    help="N > 1, all-reduce form, --launch graph / auto: record the level-group reductions INSIDE the captured step "
                         "(second branch of the graph: communication stream -> RCCL's stream, beside the scatter of the later groups) "
                         "instead of one message behind the backward")
    ap.add_argument("--split-loss", action="store_true", help="loss head as its own kernels after Renderer.forward (two-call form)")
    ap.add_argument("--exchange", choices=("shard", "allreduce"), default="allreduce",
                    help="N > 1: how the ranks exchange a step's gradients.  allreduce (default; BASELINE.json's step: fwd + bwd + "
                         "RCCL sum all-reduce of the flat gradient buffer, no optimizer in the step -- like for like with the N = 1 "
                         "line, launched the same way: eager with the level-group overlap, or one hipGraph replay); shard: "
                         "reduce-scatter -> Adam on this rank's 1/N of the parameters (ls2fm.dist.ShardedAdam) -> all-gather of the "
                         "updated shards: the step then CONTAINS the update (`update_in_step: true` in the line)")
    ap.add_argument("--no-shard", dest="exchange", action="store_const", const="allreduce")
    ap.add_argument("--shard", dest="exchange", action="store_const", const="shard")
    ap.add_argument("--shard-groups", type=int, default=1,
                    help="N > 1, --exchange shard: level groups of the PIPELINED exchange (per group, issued from inside the backward: "
                         "reduce-scatter -> Adam on this rank's slice -> all-gather); 1 (default: the form that measured faster on a one-rank RCCL group, "
                         "DESIGN 6) = one reduce-scatter / one all-gather around the update")
    ap.add_argument("--with-update", action="store_true",
                    help="N = 1: put the optimizer update (FusedAdam, schedule on the device) INSIDE the timed step, launched eagerly -- "
                         "the like-for-like single-GPU point of a scaling curve whose N > 1 points (sharded exchange) contain their update")
    ap.add_argument("--inference", action="store_true",
                    help="time the forward-only render under no_grad (the eval path: Camera.render_img_by_slices renders an image "
                         "in rand_rays chunks, pipelines/Camera.py:274-311) instead of the training step")
    args = ap.parse_args()
    preset = CONFIGS[args.config]
    args.rays = preset["rays"] if args.rays is None else args.rays
    args.samples = preset["samples"] if args.samples is None else args.samples
    args.dataset = preset["dataset"] if args.dataset is None else args.dataset
    args.single_field = (not preset["dual"]) if args.single_field is None else args.single_field

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.dataset, not args.single_field, args.samples, args.rays)))
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU, on the CPUs of the GPU's NUMA node -- BEFORE the HIP runtime initialises (ls2fm/numa.py): what a launcher's
    # `numactl --cpunodebind` does.  (Until round 6 `shade_bwd` depended on it -- its waves read the dispatch packet in host memory:
    # 123 us from the GPU's node, 135 from the other socket; now 115 us from either, LS2FM_NUMA_BIND=0 measures the same.)
    from ls2fm.numa import bind_to_gpu_numa_node
    numa_node = bind_to_gpu_numa_node(local_rank if os.environ.get("LS2FM_BENCH_BACKEND", "nccl") == "nccl" else 0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product has no CPU path); the CPU figure is the "
                         "`cpu_baseline` leg of the GPU run")
    # LS2FM_BENCH_BACKEND=gloo (tests of the N > 1 wiring on a one-GPU box: RCCL refuses two ranks per device, gloo carries
    # device tensors): every rank then shares GPU 0 -- the figures of such a run mean nothing, its control flow is the real one
    backend = os.environ.get("LS2FM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ls2fm import _lib, fused
    from ls2fm.dist import GradAllReducer, enable_table_overlap
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer

    lib = _lib.load()
    dual = not args.single_field
    opt = make_options(args.dataset, device=str(dev), dual_field=dual, sample_intvs=args.samples)
    torch.manual_seed(0)
    sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
    randomize([sdf, rad], seed=0)                         # identical replicas on every rank
    s = float(opt.data.bound_max[0])
    center, ray = synthetic_rays(args.rays, s, dev, seed=rank)      # each rank: its own view's rays
    assert fused.can_render(ren, opt, center, ray, sdf, rad), "fused HIP path not taken"
    params = list(sdf.parameters()) + list(rad.parameters())
    shard = multi and args.exchange == "shard" and not args.inference
    reducer = GradAllReducer(params) if (multi and not shard and not args.inference) else None
    sharded_opt = None
    if shard:
        # reduce-scatter -> Adam on this rank's shard -> all-gather: the same bytes on xGMI as the all-reduce, optimizer state
        # and update traffic / world; the step then includes the (sharded) update
        from ls2fm.dist import ShardedAdam
        sharded_opt = ShardedAdam.for_fields(sdf, rad, lr=1e-4, lr_color=1e-4, scheduled_gamma=1.0, n_groups=args.shard_groups,
                                             async_gather=args.shard_groups > 1)
        params = list(sharded_opt.params)
    local_opt = None
    if args.with_update and not multi and not args.inference:
        from ls2fm.optim import FusedAdam
        local_opt = FusedAdam([dict(params=list(sdf.parameters()), lr=1e-4), dict(params=list(rad.parameters()), lr=1e-4)],
                              scheduled_gamma=1.0)
    if args.capture_overlap:
        from ls2fm.dist import enable_capture_overlap
        enable_capture_overlap(True)
    if reducer is not None and not args.no_overlap:
        # the ~105 MB gradient exchange is as long as the step on 7 xGMI links and all of it comes out of the backward's last
        # kernels: scatter the levels in 4 groups and all-reduce a group's table slices while the next ones are scattered
        enable_table_overlap(sdf, rad, n_groups=args.overlap_groups)

    if sharded_opt is None and local_opt is None:
        # the timed loop never changes the tables (no optimizer in the step): a captured step needs no rebuild of the
        # interleaved table copy inside the graph
        fused.trust_mirror_in_capture(sdf, rad)
    from ls2fm.losses import RenderLossHead
    from ls2fm.graph import CapturedStep
    # same three terms and weights as loss_head().  N > 1: every rank holds the same number of rays and no masks, so the global
    # counts are the local ones times the world size ("uniform": no collective between forward and backward); the summed
    # (all-reduced) gradients are then those of the global means
    head = RenderLossHead(dev, w_rgb=3.0, w_eikonal=2.0, w_dc=0.0, global_counts="uniform")
    rgb_gt = torch.full((1, args.rays, 3), 0.5, device=dev)
    depth_ref = torch.zeros(1, args.rays, device=dev)
    one = torch.ones((), device=dev)              # d loss / d loss (what loss.backward() would allocate and fill every step)

    def render_step():
        """Renderer.forward -> loss head -> backward: every parameter's .grad is (re)written"""
        if sharded_opt is not None:
            sharded_opt.wait_params()       # the previous step's all-gathers (device-side wait) before the first parameter read
        if args.inference:
            with torch.no_grad():
                return ren.forward(opt, center, ray, sdf, rad)["rgb"]
        for p in params:
            p.grad = None
        if args.torch_loss:
            loss = loss_head(ren.forward(opt, center, ray, sdf, rad))
        elif args.split_loss:           # Renderer.forward, then the loss head as its own two kernels
            loss = head.terms(ren.forward(opt, center, ray, sdf, rad), rgb_gt, d_points=depth_ref)[1]
        else:                           # the loss head inside the render (forward epilogue / backward prologue)
            loss = ren.forward_with_loss(opt, center, ray, sdf, rad, head, rgb_gt, d_points=depth_ref)[1]["all"]
        loss.backward(gradient=one)
        return loss

    mode = "graph" if args.graph else args.launch
    if shard and mode != "graph":
        mode = "eager"                  # sharded exchange: eager unless asked.  `--launch graph` captures the WHOLE sharded step --
                                        # render, the RCCL collectives, the sharded update -- into one hipGraph (measured with a
                                        # one-rank RCCL group: 0.79 -> 0.74 ms/step)
    if local_opt is not None:
        mode = "eager"                  # the like-for-like N = 1 point of the eager sharded lines
    if multi and backend != "nccl":
        mode = "eager"                  # (gloo moves device tensors through the host: nothing a hipGraph could record)
    # all-reduce form (the default): launched like the N = 1 line -- `auto` probes eager (level-group reductions issued from
    # inside the backward, beside the scatter of the later groups) against ONE hipGraph replay of render + all-reduce of the flat
    # gradient buffer (RCCL's kernels are recorded into the graph; no overlap inside a capture, ls2fm.fused)
    # every step -- eager or captured -- runs on ONE non-default stream: autograd's gradient accumulators stay tied to
    # the stream of their first backward, and mixing streams costs synchronisations (and breaks captures)
    s_main = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(s_main)
    captured = None
    use_graph = False

    def whole_step():
        """what a graph captures: the render step and the gradient exchange behind it (sharded form: + the update)"""
        render_step()
        if sharded_opt is not None:
            sharded_opt.step()
        if reducer is not None:
            reducer.all_reduce()

    def step(eager=False):
        if use_graph and not eager:
            captured.replay()
        else:
            whole_step()
        if local_opt is not None:
            local_opt.step()

    launch_probe = None
    if mode == "graph":
        captured = CapturedStep(whole_step, params, stream=s_main)
        use_graph = True
    if mode == "auto":                      # untimed probe (part of the warm-up): pick the faster launch mode on this host
        def probe(graph, n=40):
            nonlocal use_graph
            use_graph = graph
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        t_eager = probe(False)
        # N > 1: the capture records RCCL's collectives into the graph -- exercised on a one-rank RCCL group only so far; if it
        # fails on a real multi-GPU node the line falls back to eager launches (every rank takes the same decision) and says so
        capture_error = None
        try:
            captured = CapturedStep(whole_step, params, stream=s_main)
        except Exception as e:                                   # noqa: BLE001
            if not multi:
                raise
            capture_error = f"{type(e).__name__}: {e}"[:300]
        if multi:
            ok = torch.tensor([0.0 if capture_error else 1.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() < 1.0:
                captured = None
                capture_error = capture_error or "capture failed on another rank"
        t_graph = probe(True) if captured is not None else float("inf")
        launch_probe = {"eager_ms_per_step": t_eager * 1e3, "graph_ms_per_step": None if captured is None else t_graph * 1e3}
        if capture_error:
            launch_probe["capture_error"] = capture_error
        # N > 1, all-reduce form: a THIRD candidate -- the captured step with the level-group reductions on a second branch of
        # the graph (ls2fm.dist.enable_capture_overlap: the first group's all-reduce beside the scatter of the later groups).  On a
        # one-rank group it can only lose (two scatter launch pairs, nothing to hide: 0.49 against 0.44 ms), so it is probed on
        # real multi-GPU runs only (LS2FM_BENCH_PROBE_OVERLAP=1 forces it); every rank takes the decision from the same MAX-reduced times.
        if (captured is not None and reducer is not None and not args.no_overlap and not args.capture_overlap
                and (world > 1 or os.environ.get("LS2FM_BENCH_PROBE_OVERLAP", "0") == "1")):
            from ls2fm.dist import enable_capture_overlap
            plain, overlap_error = captured, None
            enable_capture_overlap(True)
            try:
                captured = CapturedStep(whole_step, params, stream=s_main)
            except Exception as e:                               # noqa: BLE001
                overlap_error = f"{type(e).__name__}: {e}"[:300]
            ok = torch.tensor([0.0 if overlap_error else 1.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            t_overlap = probe(True) if ok.item() >= 1.0 else float("inf")
            times = torch.tensor([t_graph, t_overlap], device=dev, dtype=torch.float64)
            dist.all_reduce(times, op=dist.ReduceOp.MAX)
            t_graph_all, t_overlap_all = times.tolist()
            launch_probe["graph_level_group_overlap_ms_per_step"] = None if t_overlap == float("inf") else t_overlap * 1e3
            if overlap_error:
                launch_probe["overlap_capture_error"] = overlap_error
            if t_overlap_all < t_graph_all:
                t_graph = t_overlap
                args.capture_overlap = True                      # (the line's exchange.form says which graph ran)
            else:
                captured = plain
                enable_capture_overlap(False)
        use_graph = captured is not None and not (t_eager < 0.98 * t_graph)      # a tie goes to the replay: its step time does not depend on the host
        # (round 6: a 1 % band picked eager launches at 8192 rays -- 1 - 1.6 % faster in the probe and in same-session A/Bs on a fast host --
        # and the timed blocks then came out SLOWER than the replay of the previous collection: 2.882 against 2.8605 ms.  2 % stays.)
        if rank == 0:
            print(f"[bench] launch probe: eager {t_eager * 1e3:.3f} ms/step, hipGraph {t_graph * 1e3:.3f} ms/step"
                  + (f" (capture failed: {capture_error})" if capture_error else ""), file=sys.stderr)
        if world > 1:                       # every rank must take the same path
            flag = torch.tensor([1.0 if use_graph else 0.0], device=dev)
            dist.all_reduce(flag)
            use_graph = bool(flag.item() * 2 > world)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # timed region: EXACTLY K steps between barrier + synchronize on both sides.  That block is repeated until ~1 s of device
    # time has been sampled (a driver run with --steps 20 would otherwise time 12 ms) and the MEDIAN block is reported; every
    # block is a complete measurement under the contract, min / max are in the line.
    def timed_block():
        barrier()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        return time.perf_counter() - t_start
    blocks = [timed_block()]
    n_blocks = max(1, min(100, int(1.0 / max(blocks[0], 1e-4) + 0.999)))
    if multi:                                # every rank must run the same number of blocks
        nb = torch.tensor([n_blocks], device=dev)
        dist.all_reduce(nb, op=dist.ReduceOp.MAX)
        n_blocks = int(nb.item())
    while len(blocks) < n_blocks:
        blocks.append(timed_block())
    if multi:                                # max over ranks, block by block
        tb = torch.tensor(blocks, device=dev, dtype=torch.float64)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        blocks = tb.tolist()
    dt = sorted(blocks)[len(blocks) // 2]
    # N > 1: what the gradient exchange costs, measured outside the timed region -- the same K steps WITHOUT exchange and update
    # (compute only), so that a scaling run records exposed exchange time next to the bytes it moves.  Not
    # for the level-group overlap, whose reductions are issued from inside the backward.
    exchange = None
    if multi and not args.inference:
        flat_bytes = 4 * sum((p.numel() + 3) // 4 * 4 for p in params)
        exchange = {"form": (f"reduce-scatter + sharded Adam + all-gather, pipelined over {sharded_opt.n_groups} level groups from inside "
                             "the backward" if shard and sharded_opt.n_groups > 1 else "reduce-scatter + sharded Adam + all-gather") if shard else
                            ("all-reduce" if args.no_overlap else f"all-reduce, {args.overlap_groups} level groups overlapped with the backward"),
                    "gradient_bytes": flat_bytes, "bytes_on_wire_per_gpu": 2 * (world - 1) / world * flat_bytes}
        if not shard:
            if use_graph:
                exchange["form"] = (f"all-reduce inside the captured step, {args.overlap_groups} level groups on a second branch of the graph "
                                    "(beside the scatter of the later groups)" if args.capture_overlap and not args.no_overlap else
                                    "all-reduce of the flat gradient buffer inside the captured step")
        # compute only = the same K steps of fwd + bwd with NO exchange and no update, launched eagerly, on EVERY N > 1 line
        # (the first real scaling run separates exchange from compute whatever form it times).  The level-group
        # reductions a backward would launch itself are switched off for these steps.
        tabs = [p for p in params if hasattr(p, "_ls2fm_overlap_groups")]
        held = [(p, p._ls2fm_overlap_groups) for p in tabs]
        for p in tabs:
            p._ls2fm_overlap_groups = 0
        hook_held = [(p, p._ls2fm_group_exchange) for p in params if getattr(p, "_ls2fm_group_exchange", None) is not None]
        for p, _ in hook_held:
            p._ls2fm_group_exchange = None
        try:
            # launched like the timed steps: a hipGraph of fwd + bwd alone when those were replays
            compute_step = render_step
            if use_graph:
                try:
                    compute_step = CapturedStep(render_step, params, stream=s_main).replay
                except Exception:                                # noqa: BLE001  (the line then says "eager")
                    compute_step = render_step
            for _ in range(3):
                compute_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                compute_step()
            barrier()
        finally:
            for p, v in held:
                p._ls2fm_overlap_groups = v
            for p, v in hook_held:
                p._ls2fm_group_exchange = v
        t_compute = torch.tensor([(time.perf_counter() - t0) / args.steps], device=dev, dtype=torch.float64)
        dist.all_reduce(t_compute, op=dist.ReduceOp.MAX)
        exchange["compute_only_ms_per_step"] = float(t_compute.item()) * 1e3
        exchange["compute_only_launch"] = "eager" if compute_step is render_step else "hipGraph replay"
        exchange["exchange_and_update_ms_per_step"] = dt / args.steps * 1e3 - exchange["compute_only_ms_per_step"]
    # per-kernel device times (roofline): the same K steps launched eagerly with the library's HIP-event profiler on
    # (events cannot be read back from inside a graph replay); also gives the eager step time
    lib.ls2fm_profile_reset()
    lib.ls2fm_profile_enable(1)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step(eager=True)
    barrier()
    dt_eager = time.perf_counter() - t1
    lib.ls2fm_profile_enable(0)
    has_prof = True

    ms_per_step = dt / args.steps * 1e3
    value = args.rays * world * args.steps / dt
    roofline = None
    if has_prof and rank == 0:
        from ls2fm.profile import dominant_kernel_roofline
        roofline = dominant_kernel_roofline(lib, n_points=args.rays * args.samples, dual=dual,
                                            hbm_peak_gbs=HBM_PEAK_GBS, f32_peak_tflops=F32_PEAK_TFLOPS)
        # HBM-side bytes of that kernel: PMC counters cannot be read from inside the process; the committed counter summary of
        # this same command (profiles/, collected per MI355X_MICROARCH.md: separate --pmc passes) is QUOTED when the workload
        # is the default one -- with the commit it was measured at, so a stale figure is recognisable
        pmc = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json")) if os.path.exists(q)), "")
        if roofline and os.path.exists(pmc) and (args.rays, args.samples, args.dataset, dual) == (1024, 128, "ETH3D", True):
            doc = json.load(open(pmc))
            if roofline["kernel"].startswith("scatter_pair") and "scatter_fill" in doc and "slab_accumulate" in doc:
                rec = {k: doc["scatter_fill"][k] + doc["slab_accumulate"][k] + doc.get("slab_combine", {}).get(k, 0.0)     # the
                       for k in ("fetch", "write")}                   # accumulate span covers its combine launch (split slabs)
            else:
                rec = doc.get(roofline["kernel"])
            if rec:
                roofline["traffic"] = rec["fetch"] + rec["write"]
                roofline["traffic_source"] = (f"quoted from profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                              f"passes at commit {doc.get('_commit', '?')}; FETCH_SIZE doubled per the guide)")
        if roofline:
            # SURVEY 8d's whole-step figures: algorithmic table bytes (gather + scatter, both grids) and dense FLOPs of one step
            # over the measured step time
            n_pts = args.rays * args.samples
            bytes_per_pt = (4 if dual else 2) * 1024 + 16
            flops_per_pt = 115e3 if dual else 75e3
            step_s = dt / args.steps
            roofline["whole_step"] = {
                "hbm": {"bytes_per_ray": bytes_per_pt * args.samples, "achieved": n_pts * bytes_per_pt / step_s / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": n_pts * bytes_per_pt / step_s / 1e9 / HBM_PEAK_GBS},
                "mfma": {"flops_per_ray": flops_per_pt * args.samples, "achieved": n_pts * flops_per_pt / step_s / 1e12,
                         "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": n_pts * flops_per_pt / step_s / 1e12 / F32_PEAK_TFLOPS,
                         # SURVEY 8d's per-sample figure counts the UN-collapsed radiance decoder; the kernels execute the collapsed
                         # affine map (and the fused weight gradients): the fraction by the FLOPs actually executed
                         "useful_flops_per_step": roofline.get("useful_flops_per_step"),
                         "useful_achieved": roofline.get("useful_flops_per_step", 0.0) / step_s / 1e12,
                         "useful_frac": roofline.get("useful_flops_per_step", 0.0) / step_s / 1e12 / F32_PEAK_TFLOPS}}
    if args.inference and roofline:
        roofline.pop("whole_step", None)        # (the whole-step figures are the training step's)
    out = {
        "metric": "rendered rays/sec (forward only, no_grad)" if args.inference else "rendered rays/sec (fwd+bwd)", "value": value, "unit": "rays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "timed_blocks": len(blocks), "ms_per_step_min": min(blocks) / args.steps * 1e3, "ms_per_step_max": max(blocks) / args.steps * 1e3,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "launch": ("hipGraph replay of the whole step" if use_graph else "eager") + (" (auto)" if mode == "auto" else ""),
        "eager_profiled_ms_per_step": dt_eager / args.steps * 1e3,
        "launch_probe": launch_probe,           # untimed probe of both launch forms (auto mode): eager and hipGraph ms/step
        "config": {"workload": f"{args.config}: {args.dataset} bounds, {args.rays} rays x {args.samples} samples per GPU, "
                               f"{'dual' if dual else 'single'} field, L16/F2/T19 hash grid, fwd+loss+bwd"
                               + ((", RCCL reduce-scatter + sharded Adam + all-gather" if shard else ", RCCL grad all-reduce") if multi else ""),
                   "rays_per_gpu": args.rays, "samples_per_ray": args.samples, "dual_field": dual,
                   "parallelism": f"dp{world} (rays sharded by view)"},
        "roofline": roofline,
        "host_numa_node": numa_node,         # the GPU's NUMA node this process was bound to before the runtime started (None: not bound)
    }
    if exchange is not None:
        out["exchange"] = exchange
    # is an optimizer update part of the timed step?  (N > 1 with the sharded exchange: yes; N = 1: only with --with-update --
    # compare a scaling curve's points like for like)
    out["update_in_step"] = bool(shard or local_opt is not None)
    if dual and not multi and not args.inference:
        # The dual-field render reads the ENTRY-INTERLEAVED copy of the two tables, which FusedAdam keeps current inside its own
        # pass.  The default line times fwd + bwd only, so that upkeep is outside the timed region (with --with-update it is
        # inside); its cost is measured here, after the timed region: the paired Adam job that also writes the copy against the
        # two plain jobs (learning rate 0: the weights do not move).
        out["mirror_upkeep"] = {"in_timed_region": local_opt is not None, "adam_pair_extra_us": mirror_upkeep_us(sdf, rad, dev)}
        # the training-relevant figure one field away: a step of a training loop also pays the copy's upkeep (in its optimizer pass)
        out["ms_per_step_incl_mirror_upkeep"] = ms_per_step + (0.0 if local_opt is not None else out["mirror_upkeep"]["adam_pair_extra_us"] * 1e-3)
    if rank == 0:
        out["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            # the CPU leg runs in a fresh process (clean OpenMP state, no GPU context) under a hard wall-clock bound, so a
            # misbehaving host can never hang the benchmark
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--dataset", args.dataset,
                   "--samples", str(args.samples), "--rays", str(args.rays)] + (["--single-field"] if args.single_field else [])
            try:
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=150)
                out["cpu_baseline"] = json.loads(res.stdout.strip().splitlines()[-1])
            except Exception as e:                                   # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": host_threads(), "kind": "port",
                                       "sample": f"CPU leg did not finish within 150 s ({type(e).__name__})"}
    if multi:
        dist.destroy_process_group()      # RCCL writes its version banner to stdout on the way: keep the JSON line last
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL's banner sits in the C stdio buffer until exit otherwise
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
