"""bench.py's N > 1 wiring, end to end, on the one GPU a test box has: two ranks launched the way the driver launches them
(torch.distributed.run, one process per rank) with LS2FM_BENCH_BACKEND=gloo -- RCCL refuses two ranks per device, gloo carries
device tensors.  The figures of such a run mean nothing (both ranks share the GPU, the all-reduce goes through the host); what
is tested is the control flow the 2/4/8-GPU runs take: process-group set-up, sharded rays, the gradient exchange -- the default
all-reduce (the metric's step: no optimizer inside; overlapped from inside the backward, or flat with --no-overlap) and (--shard) reduce-scatter -> sharded
Adam -> all-gather (monolithic, or --shard-groups 2: pipelined by level group) --,
barriers, the block-count and max-over-ranks reductions, ONE JSON line on rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("extra", [[], ["--no-overlap"], ["--shard"], ["--shard", "--shard-groups", "2"]])
def test_bench_two_ranks_one_gpu(extra):
    env = dict(os.environ, LS2FM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--rays", "256", "--samples", "32", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["launch"] == "eager"
    assert d["config"]["parallelism"].startswith("dp2")
    # the default is BASELINE.json's step: fwd + bwd + gradient all-reduce, no optimizer inside it; --shard is the opt-in form
    assert ("reduce-scatter" in d["config"]["workload"]) == ("--shard" in extra)
    assert d["update_in_step"] == ("--shard" in extra)


def test_bench_one_rank_rccl_default_form_is_the_metrics_step():
    """`--force-dist` (one-rank RCCL group: the collectives are identities, everything else is real): the default N > 1 form is
    fwd + bwd + all-reduce of the flat gradient buffer, launched like the N = 1 line -- also as ONE hipGraph replay with the
    RCCL all-reduce recorded in it -- and carries no optimizer update"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LS2FM_DIST_SINGLE="1", MASTER_PORT=str(_free_port()))
    for launch in ("eager", "graph"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "2",
                              "--rays", "256", "--samples", "32", "--no-cpu-baseline", "--launch", launch],
                             cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["update_in_step"] is False and "all-reduce" in d["exchange"]["form"]
        assert d["launch"].startswith("hipGraph" if launch == "graph" else "eager")


def test_bench_c1_and_inference_lines():
    """--config C1 (BASELINE.json configs[0] at its own shape) and the forward-only line: one JSON line each"""
    for extra, metric in ((["--config", "C1"], "rendered rays/sec (fwd+bwd)"), (["--inference", "--rays", "512", "--samples", "32"], "rendered rays/sec (forward only, no_grad)")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                              "--launch", "eager"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["metric"] == metric and d["value"] > 0 and d["roofline"] is not None
        if "C1" in extra:
            assert d["config"]["rays_per_gpu"] == 256 and d["config"]["samples_per_ray"] == 32 and d["config"]["dual_field"] is False
