"""Fused Adam for the path's parameters.

The reference's stage loops (Initialization.py:149-179, BA.py:117-182, rendering_refine.py:78-96) drive
``torch.optim.Adam`` + ``ExponentialLR`` over two 12 M-entry hash tables and 26 small tensors.  ``FusedAdam`` is a
``torch.optim.Optimizer`` with Adam's state layout (``step, exp_avg, exp_avg_sq``) and hyper-parameters, so LR schedulers
and ``state_dict`` round trips work unchanged; ``step()`` is ONE kernel launch and one pass over memory per parameter
group (csrc/adam.hip) instead of torch's multi-kernel foreach sequence.  fp32 GPU tensors only; no CPU fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """``scheduled_gamma``: Adam + ExponentialLR(gamma) with the step count and learning rate RESIDENT ON THE DEVICE
    (ls2fm_adam_step_scheduled): ``step()`` then takes nothing from the host, so a whole optimisation step -- render, loss,
    backward, update, schedule -- can be captured into ONE hipGraph and replayed (ls2fm.stage).  ``param_groups[i]["lr"]`` is
    kept in step on the host for logging / state_dict; do not attach a torch scheduler as well."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, scheduled_gamma=None):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.scheduled_gamma = None if scheduled_gamma is None else float(scheduled_gamma)
        self._sched = {}            # group index -> device float64[4] = {step, lr, gamma, (step_size, bc2_sqrt as two floats)}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        calls = {}                  # (betas, eps, weight_decay, step or None) -> [(param, grad, state, lr, schedule tensor or None)]
        idle = []                   # device schedules of groups without a gradient this step: their rate decays all the same
        for gi, group in enumerate(self.param_groups):
            steps = set()
            items = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_device(p, p.grad)
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or p.grad.is_sparse:
                    raise RuntimeError("ls2fm.optim.FusedAdam: contiguous fp32 dense parameters and gradients only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                steps.add(st["step"])
                items.append((p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), st))
            sched = None
            if self.scheduled_gamma is not None:
                if len(steps) > 1:
                    raise RuntimeError("ls2fm.optim.FusedAdam(scheduled_gamma=...): the parameters of a group must step together")
                if gi not in self._sched:
                    first = next(iter(steps)) - 1 if steps else 0
                    dev = group["params"][0].device
                    self._sched[gi] = torch.tensor([float(first), float(group["lr"]), self.scheduled_gamma, 0.0], device=dev,
                                                   dtype=torch.float64)
                sched = self._sched[gi]
            for p, g, st in items:
                key = (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]), float(group["weight_decay"]),
                       None if sched is not None else int(st["step"]))
                calls.setdefault(key, []).append((p, g, st, float(group["lr"]), sched))
            if sched is not None:
                # ExponentialLR decays EVERY group at every scheduler step, with or without gradients (a group whose tensors all
                # have grad None this step keeps its Adam step count -- torch skips such tensors -- but not its rate)
                if not items:
                    idle.append(sched)
                group["lr"] = float(group["lr"]) * self.scheduled_gamma        # host mirror of the device schedule
        # ONE launch per distinct (betas, eps, weight decay[, step]) -- normally one for the whole optimizer, whatever the number
        # of parameter groups: per-tensor learning rates / schedules ride in the call (ls2fm_adam_step_multi)
        for (b1, b2, eps, wd, step), items in calls.items():
            n = len(items)
            ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])          # noqa: E731
            mirrors = [self._mirror_of(p) for p, _, _, _, _ in items]
            any_mirror = any(m[0] is not None for m in mirrors)
            _lib.check(lib.ls2fm_adam_step_multi(
                n, ptrs([it[0] for it in items]), ptrs([it[1] for it in items]), ptrs([it[2]["exp_avg"] for it in items]),
                ptrs([it[2]["exp_avg_sq"] for it in items]), (ctypes.c_int64 * n)(*[it[0].numel() for it in items]),
                (ctypes.c_void_p * n)(*[m[1] for m in mirrors]) if any_mirror else None,
                (ctypes.c_float * n)(*[it[3] for it in items]),
                (ctypes.c_void_p * n)(*[None if it[4] is None else it[4].data_ptr() for it in items]) if step is None else None,
                b1, b2, eps, wd, 0 if step is None else step, _lib.stream_ptr()), "ls2fm_adam_step_multi")
            for (p, _, _, _, _), (rec, _) in zip(items, mirrors):
                # the kernel wrote through raw pointers: tell autograd (and every cache keyed on Tensor._version) ...
                torch.autograd.graph.increment_version(p)
                if rec is not None:             # ... and the interleaved table copy, which already holds the new values
                    rec[0].written_by_optimizer(rec[1], p)
        if idle:
            _lib.check(lib.ls2fm_adam_sched_decay(len(idle), (ctypes.c_void_p * len(idle))(*[t.data_ptr() for t in idle]),
                                                  _lib.stream_ptr()), "ls2fm_adam_sched_decay")
        return loss

    @staticmethod
    def _mirror_of(p):
        """(record, destination pointer) when `p` is a hash table whose entry-interleaved copy the render reads (ls2fm.fused)"""
        rec = getattr(p, "_ls2fm_mirror", None)
        if rec is None or rec[0].table is None or rec[0].table.numel() != 2 * p.numel() or rec[0].ptrs[rec[1]] != p.data_ptr():
            return None, None
        return rec, rec[0].table.data_ptr() + 8 * rec[1]

    def replayed(self, n=1):
        """a captured step containing this optimizer's update was replayed n times: advance the host mirrors (state['step'],
        param_groups[i]['lr']) the device schedule already advanced"""
        for group in self.param_groups:
            for p in group["params"]:
                if p in self.state and self.state[p]:
                    self.state[p]["step"] = int(self.state[p]["step"]) + n
                    # the captured kernel rewrote the parameter through its raw pointer: tell autograd and every cache
                    # keyed on Tensor._version (the interleaved table copy) -- as step() does
                    torch.autograd.graph.increment_version(p)
                    rec, _ = self._mirror_of(p)
                    if rec is not None:         # the captured mirrored update rewrote the interleaved copy as well
                        rec[0].written_by_optimizer(rec[1], p)
            if self.scheduled_gamma is not None:
                group["lr"] = float(group["lr"]) * self.scheduled_gamma ** n

    def load_state_dict(self, state_dict):
        """IN PLACE: a captured step (ls2fm.graph.CapturedStep, RenderStage(capture=True)) has the device addresses of the moments
        and of the device-resident schedules baked into its hipGraph.  The loaded values are therefore copied INTO the tensors this
        optimizer already owns (moments, `_sched`), which a later replay then reads -- torch's loader alone would install new
        tensors and leave the graph reading (and writing 32 bytes of schedule into) memory that was handed back to the allocator.
        The device schedule is derived state: reseeded here from the loaded `step` and `lr` of its group."""
        held = {p: dict(st) for p, st in self.state.items() if st}
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for p, old in held.items():
                st = self.state.get(p)
                if not st:
                    # (round-4 advisor) live state, nothing in the checkpoint: the reference's loader would drop the entry and Adam
                    # restart it from zero moments -- the same here, but IN the tensors a captured graph may hold: zeroed, kept
                    for k in ("exp_avg", "exp_avg_sq"):
                        if torch.is_tensor(old.get(k)):
                            old[k].zero_()
                    old["step"] = 0
                    self.state[p] = old
                    continue
                for k in ("exp_avg", "exp_avg_sq"):
                    if torch.is_tensor(old.get(k)) and torch.is_tensor(st.get(k)):
                        if old[k].shape != st[k].shape:
                            raise RuntimeError(f"ls2fm.optim.FusedAdam.load_state_dict: '{k}' of a parameter has shape "
                                               f"{tuple(st[k].shape)} in the checkpoint, {tuple(old[k].shape)} here")
                        old[k].copy_(st[k])
                        st[k] = old[k]
                st["step"] = int(st["step"])
            for gi, t in self._sched.items():
                if gi >= len(self.param_groups):
                    continue
                group = self.param_groups[gi]
                steps = {int(self.state[p]["step"]) for p in group["params"] if p in self.state and self.state[p]}
                if len(steps) > 1:
                    raise RuntimeError("ls2fm.optim.FusedAdam(scheduled_gamma=...): the parameters of a group must step together")
                t.copy_(torch.tensor([float(steps.pop()) if steps else 0.0, float(group["lr"]), self.scheduled_gamma, 0.0],
                                     dtype=torch.float64))
