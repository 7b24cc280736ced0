"""HIP-graph capture of a whole optimisation step.

The reference's stage loops (Initialization.py:149-179, BA.py:117-182, rendering_refine.py:78-96) run
``Renderer.forward -> losses -> loss.backward()`` eagerly: at 1024 rays x 128 samples the device work is ~1 ms and the
Python / launch overhead between the ~15 kernels of the fused path is of the same order.  ``CapturedStep`` records the
step once into a hipGraph (the library's internal fork/join onto its side stream is captured as graph branches) and
replays it with one launch.

    step = CapturedStep(lambda: loss_fn(renderer.forward(opt, center, ray, sdf, rad)).backward() ..., params=all_parameters)
    for it in range(n): new rays -> center.copy_(...), ray.copy_(...); step.replay(); optimizer.step()

Contract (the usual one for graphs): the closure reads its inputs from tensors that keep their address (update them in
place between replays), performs no host synchronisation (`.item()`, data-dependent Python control flow), and the
tensors it produces -- including the parameters' ``.grad`` -- are rewritten in place by every replay.
"""
from __future__ import annotations

import torch


class CapturedStep:
    def __init__(self, fn, params=None, warmup: int = 3, stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("ls2fm.graph.CapturedStep needs the GPU (no CPU path)")
        self.fn = fn
        # `stream`: capture (and warm up) on this non-default stream.  Autograd ties every parameter's gradient
        # accumulator to the stream of its first backward and synchronises with it in later backwards: if eager steps of
        # the same parameters run too, run them on this very stream (torch.cuda.stream(stream)) -- mixing in the legacy
        # default stream breaks the capture and slows the eager steps (0.73 -> 1.0 ms here).
        side = stream if stream is not None else torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # No cyclic garbage collection between here and the end of the capture: a collection that fires INSIDE the capture (they
        # are triggered by allocation counts, from whichever thread is running -- autograd's worker included) can destroy device
        # objects of earlier, already unreachable steps (graphs, events, pool memory), which the runtime does not allow while a
        # stream is capturing: the process aborts.  Seen once in ~10 full test runs, in the backward of a loop's capture.
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            with torch.cuda.stream(side):                 # warm-up off the default stream: allocator pools, lazy library state
                for _ in range(warmup):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gc.collect()
            self.graph = torch.cuda.CUDAGraph()
            # A process group's watchdog THREAD polls the events of the collectives issued so far (hipEventQuery, every ~100 ms
            # until it has seen each one complete).  Under the default capture mode ("global") such a call from ANY thread while
            # this one captures is an error -- it invalidates the capture and the watchdog aborts the process ("operation not
            # permitted when stream is capturing": seen twice in ~25 full GPU test runs, in a loop's capture behind eager
            # warm-up steps with a one-rank RCCL group; an N > 1 bench would meet it the same way).  So: let the watchdog retire
            # what the warm-up issued (everything has completed: the device is idle), and capture in "thread_local" mode --
            # only THIS thread's unsafe calls are errors; kernels other threads launch on the capturing stream (autograd's
            # workers) are recorded as before.
            mode = "global"
            dist = getattr(torch, "distributed", None)
            if dist is not None and dist.is_available() and dist.is_initialized():
                import time
                time.sleep(0.35)
                mode = "thread_local"
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=mode):
                self.outputs = fn()
        finally:
            if gc_was_on:
                gc.enable()
        # the gradients the capture produced live in the graph's memory pool; `params` lets replay() re-bind them after
        # eager code replaced p.grad in between
        self._grads = [(p, p.grad) for p in params] if params is not None else []

    def replay(self):
        self.graph.replay()
        for p, g in self._grads:
            p.grad = g
        return self.outputs
