"""`FusedAdam`: torch.optim.Adam whose `step()` is ONE kernel launch (`lic_adam_run`, include/lic.h) over every
parameter instead of nine multi-tensor launches.

A subclass of torch.optim.Adam: same constructor, same `state_dict()` (`step`, `exp_avg`, `exp_avg_sq` per
parameter, created by torch's own `_init_group`), so checkpoints move freely between the two -- the reference's
`Trainer` takes whatever optimizer it is given (Trainer.py:11-16) and the notebook builds `torch.optim.Adam`
(Main.ipynb).  The arithmetic is torch's non-amsgrad Adam with L2 weight decay, bias corrections computed on
the host from the step count (torch's default, non-capturable path); anything else -- amsgrad, maximize,
capturable / differentiable, non-fp32 or CPU parameters, parameters without gradients, step counts that differ
between parameters -- falls through to torch's implementation for that call.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import functional as F_


class FusedAdam(torch.optim.Adam):
    MAX_TENSORS = 448   # per parameter group (include/lic.h: lic_adam_run)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._tables = {}   # group index -> (key, device job table, njobs, blocks, ctypes array of gradient pointers)
        self._fast = None   # the last planned step, re-used while nothing it depends on changed
        self._t0, self._lazy = 0.0, 0   # step count the plan started from / fast steps not yet written to the state
        self.fused_steps = 0

    def _eligible(self, group, params, grads):
        if group["amsgrad"] or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if isinstance(group["lr"], torch.Tensor) or not params or len(params) > self.MAX_TENSORS:
            return False   # (lic_adam_run's kernel-argument block holds MAX_TENSORS gradient addresses)
        dev = params[0].device
        for p, g in zip(params, grads):
            if p.device != dev or not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or \
                    g.is_sparse or not p.is_contiguous() or not g.is_contiguous():
                return False
        return dev.index == torch.cuda.current_device()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # Fast path (the host side of the step is what bounds the bf16 configurations: planning a launch through
        # torch's `_init_group` and re-deriving the job-table key cost 0.5 ms of a 4.4 ms step): once a step has
        # been planned, the next one only checks that the groups' parameter lists and hyper-parameters are the very
        # objects / values it planned for and that every gradient is there, then gathers the gradient addresses.
        fast = self._fast
        if fast is not None and len(fast) == len(self.param_groups):
            ok = True
            for (group, plist, params, hyper, (tab, _, _)), g in zip(fast, self.param_groups):
                if g is not group or g["params"] is not plist or len(plist) != len(params) or \
                        (g["lr"], g["betas"], g["eps"], g["weight_decay"], g["amsgrad"], g.get("maximize"),
                         g.get("capturable"), g.get("differentiable")) != hyper:
                    ok = False
                    break
                for p, k in zip(params, tab[0]):
                    gr = p.grad
                    if gr is None or gr.dtype != torch.float32 or not gr.is_contiguous() or p.data_ptr() != k[0]:
                        ok = False   # (a missing gradient, or parameter storage that moved: plan again)
                        break
                if not ok:
                    break
            if ok:
                lib = L.load()
                for group, plist, params, hyper, (tab, moments, steps) in fast:
                    _, dev_tab, njobs, blocks, gptrs = tab
                    for i, p in enumerate(params):
                        gptrs[i] = p.grad.data_ptr()
                    t = self._t0 + self._lazy + 1.0
                    beta1, beta2 = group["betas"]
                    L.check(lib.lic_adam_run(C.c_void_p(dev_tab.data_ptr()), njobs, blocks, gptrs, float(group["lr"]),
                                             float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]),
                                             1.0 - beta1 ** t, 1.0 - beta2 ** t, F_._stream()), "lic_adam_run")
                    torch.autograd.graph.increment_version(moments)
                # the per-parameter `step` tensors of the state (CPU scalars, one per parameter: bumping them is ~240
                # small ATen calls per step) are brought up to date lazily: _flush_steps() before anything reads them
                self._lazy += 1
                self.fused_steps += 1
                return loss
        self._flush_steps()
        self._fast = None
        plans = []
        for gi, group in enumerate(self.param_groups):
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if len(params) != len([p for p in group["params"]]) or not self._eligible(group, params, grads) or \
                    not all(m.is_contiguous() and v.is_contiguous() for m, v in zip(exp_avgs, exp_avg_sqs)) or \
                    len({float(s) for s in steps}) != 1:
                return self._fallback(loss)
            plans.append((gi, group, params, grads, exp_avgs, exp_avg_sqs, steps))
        lib = L.load()
        fast = []
        for gi, group, params, grads, exp_avgs, exp_avg_sqs, steps in plans:
            key = tuple((p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, m, v in zip(params, exp_avgs, exp_avg_sqs))
            tab = self._tables.get(gi)
            if tab is None or tab[0] != key:
                arr = (L.AdamJob * len(params))()
                for j, (p, m, v) in zip(arr, zip(params, exp_avgs, exp_avg_sqs)):
                    j.p, j.m, j.v, j.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                blocks = lib.lic_adam_plan(arr, len(params))
                if blocks <= 0:
                    raise L.LicError(f"lic_adam_plan failed: {blocks}")
                dev_tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(params[0].device)
                tab = self._tables[gi] = (key, dev_tab, len(params), int(blocks), (C.c_void_p * len(params))())
            _, dev_tab, njobs, blocks, gptrs = tab
            for i, g in enumerate(grads):       # gradients are fresh tensors every step: their addresses go along
                gptrs[i] = g.data_ptr()         # as kernel arguments (copied at launch)
            t = float(steps[0]) + 1.0
            beta1, beta2 = group["betas"]
            L.check(lib.lic_adam_run(C.c_void_p(dev_tab.data_ptr()), njobs, blocks, gptrs, float(group["lr"]), float(beta1),
                                     float(beta2), float(group["eps"]), float(group["weight_decay"]),
                                     1.0 - beta1 ** t, 1.0 - beta2 ** t, F_._stream()), "lic_adam_run")
            for s in steps:     # (only once the launch went out: a refused launch leaves the state untouched)
                s += 1
            # the kernel wrote through raw pointers: tell autograd (and prep.StepPrep, which re-derives the packed
            # weights when a parameter's version moves) that the parameters and moments changed
            moments = params + exp_avgs + exp_avg_sqs
            torch.autograd.graph.increment_version(moments)
            hyper = (group["lr"], group["betas"], group["eps"], group["weight_decay"], group["amsgrad"],
                     group.get("maximize"), group.get("capturable"), group.get("differentiable"))
            # (the job table holds the parameter / moment addresses: valid while these very tensors are the state)
            if all(p.data_ptr() == k[0] for p, k in zip(params, tab[0])) and len(params) == len(group["params"]):
                fast.append((group, group["params"], list(params), hyper, (tab, moments, list(steps))))
        if len(fast) == len(self.param_groups) and len({float(f[4][2][0]) for f in fast}) == 1:
            self._fast = fast
            self._t0, self._lazy = float(fast[0][4][2][0]), 0
        self.fused_steps += 1
        return loss

    def _flush_steps(self):
        """apply the step counts the fast path has not yet written into the state's `step` tensors"""
        n, fast = getattr(self, "_lazy", 0), self._fast
        if n and fast is not None:
            for _, _, _, _, (_, _, steps) in fast:
                for s in steps:
                    s += n
            self._t0 += n
        self._lazy = 0

    def state_dict(self):
        self._flush_steps()
        return super().state_dict()

    def zero_grad(self, set_to_none: bool = True):
        # (torch's zero_grad walks every group through a profiler scope and foreach bookkeeping: 0.1 ms for 59 tensors)
        if set_to_none and self._fast is not None:
            for _, _, params, _, _ in self._fast:
                for p in params:
                    p.grad = None
            return
        return super().zero_grad(set_to_none)

    def load_state_dict(self, state_dict):
        self._flush_steps()
        self._fast = None          # (new state tensors: plan again)
        self._tables = {}
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        if getattr(self, "_fast", None) is not None:
            self._flush_steps()
        self._fast = None
        return super().add_param_group(param_group)

    def _fallback(self, loss):
        """torch's own update for this call (state was initialised by the same `_init_group`)"""
        self._flush_steps()
        self._fast = None
        super().step()
        return loss
