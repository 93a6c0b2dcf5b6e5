"""StepPlan: the training step (Trainer.py:78-85: forward, rd_loss, backward) captured once and replayed by
liblic_hip.so's launch plan (include/lic.h, lic_plan_*) -- the same kernels on the same two streams as the eager
step, with no Python between the launches.

Why: the bf16 configurations are host-paced.  One step is ~180 launches; the Python around them (autograd nodes,
allocations, ctypes marshalling) costs ~3.1 ms of a 3.8 ms step, the launches themselves 0.7 ms, the GPU work ~3 ms.
A HIP graph would remove the host cost, but this ROCm replays a graph with more than one branch node by node from
the host (10 ms per step) and a single-branch graph loses the decoder / latent-side overlap.  So torch captures the
step only as a RECORD (private memory pool: fixed addresses); lic_plan_create turns the record into two launch
sequences with one event per cross-stream edge, and every replay is one C call.

What stays outside the plan, in eager Python: the uniform noise of the quantiser (one generator launch into a fixed
buffer: the draw is the reference's, z first -- Models.py:57-58 -- and advances torch's generator exactly as the
eager step does, so eager and planned steps produce the same bits from the same seed) and the optimizer
(FusedAdam: one launch; its bias corrections are host arithmetic).

Limits: one process, one GPU (the gradient all-reduce of parallel.py runs from autograd hooks, which a replay does
not execute: use the eager step for data-parallel training); fixed batch shape; training mode."""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

import torch

from . import _lib as L
from . import functional as F_
from .functional import _stream


class StepPlan:
    def __init__(self, model, loss_fn: Callable, lambda_val: float, example: torch.Tensor, warmup: int = 2,
                 side_stream: Optional[torch.cuda.Stream] = None, tune: bool = False):
        """`loss_fn(model_out, x, lambda_val, sync=False)` -> dict with 'loss' (loss.rd_loss); `example`: a batch
        of the shape every later batch will have.  Runs `warmup` eager steps WITHOUT an optimizer step (they fill
        the caches: packed-weight plan, gradient buffers, allocator), then captures one."""
        if not example.is_cuda:
            raise L.LicError("StepPlan needs a CUDA batch (there is no CPU fallback)")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1:
            raise L.LicError("StepPlan replays one GPU's step; data-parallel training uses the eager step "
                             "(the gradient all-reduce runs from autograd hooks)")
        self.model, self.loss_fn, self.lam = model, loss_fn, lambda_val
        self.x = example.clone(memory_format=torch.preserve_format)
        self._plan = C.c_void_p()
        self._lib = L.load()
        dev = example.device
        self._one = torch.ones((), device=dev, dtype=torch.float32)
        gen_state = torch.cuda.get_rng_state(dev)   # (the probe and the warm-up draw noise: restored below)
        with torch.no_grad():
            probe = model.analysis_hyperprior(self.x, training=True) if hasattr(model, "analysis_hyperprior") \
                else model(self.x)
        y, z = probe["y"], probe["z"]
        Bn, Mc, hy, wy = y.shape
        _, Mz, hz, wz = z.shape
        nz, ny = Bn * hz * wz * Mz, Bn * hy * wy * Mc
        # one buffer, z's noise first: the layout (NHWC per latent) and the draw order of models.analysis_hyperprior
        self.u = torch.empty(nz + ny, device=dev, dtype=y.dtype)
        self.noise = (self.u[:nz].view(Bn, hz, wz, Mz).permute(0, 3, 1, 2),
                      self.u[nz:].view(Bn, hy, wy, Mc).permute(0, 3, 1, 2))
        del probe, y, z
        # the second stream: the model's own (high priority: the decoder chain it carries is the critical path of the
        # overlapped region -- the plan's second stream inherits the capture's fork, so it carries that chain too)
        if side_stream is None and hasattr(model, "side_stream"):
            side_stream = model.side_stream()
        self.side = side_stream if side_stream is not None else torch.cuda.Stream(device=dev)
        self.side2 = torch.cuda.Stream(device=dev)
        self._sides = (C.c_void_p * 2)(self.side.cuda_stream, self.side2.cuda_stream)
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            for _ in range(max(warmup, 1)):
                self._eager_body()
        torch.cuda.current_stream(dev).wait_stream(cap)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        self._zero_grads()
        # launches the eager step SKIPS when nothing changed must be in the record: the warm-up took no optimizer
        # step, so the per-step weight preparation (prep.StepPrep.run) would see unchanged version counters
        prep = getattr(model, "_step_prep", None)
        if prep is not None:
            prep.invalidate()
        with torch.cuda.graph(self.graph, stream=cap):
            self.out, self.results = self._body()
        torch.cuda.set_rng_state(gen_state, dev)   # (building the plan leaves torch's generator where it found it)
        rc = self._lib.lic_plan_create(C.c_void_p(self.graph.raw_cuda_graph()), C.byref(self._plan))
        if rc != 0:
            why = self._lib.lic_plan_last_error().decode()
            self._plan = C.c_void_p()
            raise L.LicError(f"lic_plan_create failed: {L._STATUS.get(rc, rc)} ({why})")
        self.tuning = None
        if tune:
            # time every operation, list-schedule onto the two streams, keep whichever schedule measures faster
            # (a few dozen replays of the captured step: same inputs, same outputs)
            r = (C.c_double * 4)()
            rc = self._lib.lic_plan_tune(self._plan, _stream(), self._sides, 2, r)
            if rc != 0:
                raise L.LicError(f"lic_plan_tune failed: {L._STATUS.get(rc, rc)} ({self._lib.lic_plan_last_error().decode()})")
            torch.cuda.synchronize(dev)
            self.tuning = {"sum_of_operations_us": round(r[0], 1), "capture_order_us": round(r[1], 1),
                           "tuned_us": round(r[2], 1), "tuned_kept": bool(r[3])}
        info = (C.c_int64 * 7)()
        L.check(self._lib.lic_plan_info(self._plan, info), "lic_plan_info")
        self.info = dict(zip(("nodes", "kernels", "memsets", "memcpys", "on_side_stream", "events", "tuned"), list(info)))
        self.replays = 0
        self.streams = 3   # 1: everything on the caller's stream, 2: one side stream (A/B, debugging)
        self._params = list(model.parameters())
        self._grads = [p.grad for p in self._params]   # the captured step's gradient tensors: every replay rewrites them

    def _zero_grads(self):
        for p in self.model.parameters():
            p.grad = None
        for e in F_.GRAD_VIEWS.values():   # (bucket slots of a one-rank reducer may go out again)
            e[2] = False

    def _body(self):
        # (recorded steps launch the decoder's and the latent side's pending reductions as soon as dL/dy is complete, on the
        # second stream: functional.flush_point -- worth ~1 % replayed, harmful eager)
        early, F_.EARLY_FLUSH = F_.EARLY_FLUSH, os.environ.get("LIC_EARLY_FLUSH", "1") != "0"
        F_.PLAN_RECORDING = True
        try:
            out = self.model(self.x, noise=self.noise)
            res = self.loss_fn(out, self.x, self.lam, sync=False)
            res["loss"].backward(self._one)   # (a resident 1.0: `backward()` alone launches a fill for it every step)
        finally:
            F_.EARLY_FLUSH = early
            F_.PLAN_RECORDING = False
        return out, res

    def _eager_body(self):
        self._zero_grads()
        self.u.uniform_()
        return self._body()

    def step(self, x: torch.Tensor):
        """forward + loss + backward of `x` (same shape as the example): gradients land in `p.grad` of every
        parameter (the tensors of the captured step: same addresses every time); returns (model_out, results), the
        captured step's output tensors, rewritten by every call."""
        if x.shape != self.x.shape:
            raise L.LicError(f"StepPlan was captured for batches of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        for p, g in zip(self._params, self._grads):   # (an optimizer.zero_grad(set_to_none=True) in between)
            if p.grad is not g:
                p.grad = g
        self.u.uniform_()
        rc = self._lib.lic_plan_replay(self._plan, _stream(), self._sides, self.streams - 1)
        if rc != 0:
            raise L.LicError(f"lic_plan_replay failed: {L._STATUS.get(rc, rc)} ({self._lib.lic_plan_last_error().decode()})")
        self.replays += 1
        return self.out, self.results

    def close(self):
        if self._plan:
            self._lib.lic_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ForwardPlan:
    """The analysis + hyperprior forward (models.analysis_hyperprior: the scope BASELINE's north star quotes its roofline
    on; also what a serving process runs before the entropy coder) captured once under no_grad and replayed by the
    library -- StepPlan without the backward pass.  `training`: with the quantiser's uniform noise (drawn per call into a
    fixed buffer, the reference's order) or with rounding."""

    def __init__(self, model, example: torch.Tensor, training: bool = True, warmup: int = 2):
        if not example.is_cuda:
            raise L.LicError("ForwardPlan needs a CUDA batch (there is no CPU fallback)")
        self.model, self.training = model, training
        self._lib = L.load()
        self._plan = C.c_void_p()
        self.x = example.clone(memory_format=torch.preserve_format)
        dev = example.device
        gen_state = torch.cuda.get_rng_state(dev)
        with torch.no_grad():
            probe = model.analysis_hyperprior(self.x, training=training)
        y, z = probe["y"], probe["z"]
        Bn, Mc, hy, wy = y.shape
        _, Mz, hz, wz = z.shape
        nz, ny = Bn * hz * wz * Mz, Bn * hy * wy * Mc
        self.u = torch.empty(nz + ny, device=dev, dtype=y.dtype)
        self.noise = (self.u[:nz].view(Bn, hz, wz, Mz).permute(0, 3, 1, 2),
                      self.u[nz:].view(Bn, hy, wy, Mc).permute(0, 3, 1, 2)) if training else None
        del probe, y, z
        self.side = model.side_stream() if hasattr(model, "side_stream") else torch.cuda.Stream(device=dev)
        self._sides = (C.c_void_p * 1)(self.side.cuda_stream)
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap), torch.no_grad():
            for _ in range(max(warmup, 1)):
                self.u.uniform_()
                model.analysis_hyperprior(self.x, training=training, noise=self.noise)
        torch.cuda.current_stream(dev).wait_stream(cap)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=cap):
            self.out = model.analysis_hyperprior(self.x, training=training, noise=self.noise)
        torch.cuda.set_rng_state(gen_state, dev)
        rc = self._lib.lic_plan_create(C.c_void_p(self.graph.raw_cuda_graph()), C.byref(self._plan))
        if rc != 0:
            why = self._lib.lic_plan_last_error().decode()
            self._plan = C.c_void_p()
            raise L.LicError(f"lic_plan_create failed: {L._STATUS.get(rc, rc)} ({why})")
        info = (C.c_int64 * 7)()
        L.check(self._lib.lic_plan_info(self._plan, info), "lic_plan_info")
        self.info = dict(zip(("nodes", "kernels", "memsets", "memcpys", "on_side_stream", "events", "tuned"), list(info)))

    def __call__(self, x: torch.Tensor):
        if x.shape != self.x.shape:
            raise L.LicError(f"ForwardPlan was captured for batches of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        if self.training:
            self.u.uniform_()
        rc = self._lib.lic_plan_replay(self._plan, _stream(), self._sides, 1)
        if rc != 0:
            raise L.LicError(f"lic_plan_replay failed: {L._STATUS.get(rc, rc)} ({self._lib.lic_plan_last_error().decode()})")
        return self.out

    def close(self):
        if self._plan:
            self._lib.lic_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
