"""Data parallelism for the hot path: one process per GPU, parameters replicated, the batch
sharded, and ONE exchange step -- an all-reduce(mean) of the gradients over RCCL/xGMI -- before
optimizer.step().  The reference has no distributed code at all: the contract
is "same gradients as one big batch" (rd_loss terms are batch means, RateDistortionLoss.py:20-27).

Gradients live in a few large flat buckets that are allocated ONCE (sized for xGMI's per-link
bandwidth, not for NVSwitch).  Every parameter's gradient is a view into its bucket, and the weight-
gradient kernels write there directly (`functional.grad_like` hands the view to lic_wgrad as its
destination: 99 % of the gradient bytes never move again); a gradient that autograd produced elsewhere
(small vectors, split views) is copied into its slot when it arrives.  A bucket's all-reduce is launched
from an autograd hook as soon as its last gradient exists, so it overlaps the rest of backward;
`finish()` waits (and, for back-ends without an averaging collective, scales each bucket with one
multiply).  No per-step concatenation, no per-step allocation.  Works with any torch.distributed
back-end ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import weakref
from typing import List

import torch
import torch.distributed as dist

from . import functional as F_


class GradientAllReducer:
    def __init__(self, params, process_group=None, bucket_mb: float = 16.0, overlap: bool = True,
                 force: bool = False, stream_groups=None, group_streams=None):
        """`force`: run the hooks and collectives even on a single rank (rehearses the data path and its
        stream interplay on a one-GPU box; the result is unchanged: sum of one, mean of one)."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()
        self.active = self.world > 1 or self.force
        # RCCL can average in the collective (gloo has no AVG: sum, then one multiply per bucket)
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.overlap = overlap
        # buckets in reverse registration order ~ the order backward produces gradients.  `stream_groups`
        # (lists of parameters) keeps a bucket inside one group: with the model's two-stream overlap the
        # decoder's gradients are produced on another HIP stream than the rest, and a mixed bucket would
        # make whichever stream closes it wait for the other one (measured: -7 % step time).
        cap = int(bucket_mb * 1024 * 1024 / 4)
        index_of = {id(p): i for i, p in enumerate(self.params)}
        if stream_groups:
            grouped = [[index_of[id(p)] for p in g if id(p) in index_of] for g in stream_groups]
            seen = {i for g in grouped for i in g}
            grouped.append([i for i in range(len(self.params)) if i not in seen])
        else:
            grouped = [list(range(len(self.params)))]
        self.buckets: List[List[int]] = []
        for members in grouped:
            cur, cur_n = [], 0
            for i in reversed(members):
                n = self.params[i].numel()
                if cur and cur_n + n > cap:
                    self.buckets.append(cur)
                    cur, cur_n = [], 0
                cur.append(i)
                cur_n += n
            if cur:
                self.buckets.append(cur)
        self._bucket_of = {}
        for b, idxs in enumerate(self.buckets):
            for i in idxs:
                self._bucket_of[i] = b
        # persistent flat buffers + one gradient view per parameter (16-byte aligned slots: the weight-
        # gradient kernels and the optimizer's vectorised loads want that)
        self._flat: List[torch.Tensor] = []
        self._views: List[torch.Tensor] = [None] * len(self.params)
        if self.active:
            for idxs in self.buckets:
                p0 = self.params[idxs[0]]
                offs, off = [], 0
                for i in idxs:
                    offs.append(off)
                    off += (self.params[i].numel() + 3) // 4 * 4
                flat = torch.zeros(off, device=p0.device, dtype=p0.dtype)
                self._flat.append(flat)
                for i, o in zip(idxs, offs):
                    p = self.params[i]
                    self._views[i] = flat[o:o + p.numel()].view(p.shape)
                    F_.GRAD_VIEWS[id(p)] = [weakref.ref(p), self._views[i], False]   # [param, slot, handed out this step]
        self._pending = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._events = [[] for _ in self.buckets]  # one per gradient: backward may run on several streams
        # with per-group streams every gradient of a bucket is accumulated on the stream that also runs
        # the closing hook, so stream order already covers it; otherwise fence each gradient with an event
        self._fence = not (stream_groups and group_streams)
        self._hooks = []
        if self.active and overlap:
            # Registering a hook creates (and pins) the parameter's AccumulateGrad node on the CURRENT
            # stream.  A group whose gradients are produced on another stream registers under that stream
            # (`group_streams`), else autograd would synchronise the two streams at every such gradient.
            stream_of = {}
            if stream_groups and group_streams:
                for g, st in zip(stream_groups, group_streams):
                    for q in g:
                        stream_of[id(q)] = st
            for i, p in enumerate(self.params):
                st = stream_of.get(id(p))
                if st is not None and p.is_cuda:
                    with torch.cuda.stream(st):
                        self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
                else:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def reset(self):
        # indices of the parameters whose gradient had to be copied into its slot during the step that just
        # ended (diagnostic: the weight-gradient kernels are expected to write theirs in place)
        self.copied_last_step = getattr(self, "_copied", [])
        self._copied = []
        for p in self.params:   # functional.grad_like may hand every slot out again
            e = F_.GRAD_VIEWS.get(id(p))
            if e is not None and e[0]() is p:
                e[2] = False
        for b, idxs in enumerate(self.buckets):
            self._pending[b] = len(idxs)
            self._work[b] = None
            self._events[b] = []

    def _make_hook(self, i):
        def hook(_p):
            b = self._bucket_of[i]
            self._adopt(i)
            if self._fence and _p.grad is not None and _p.grad.is_cuda:
                # the model overlaps its decoder and latent branches on two HIP streams, so a bucket's
                # gradients can come from different streams: fence each one where it was produced
                ev = torch.cuda.Event()
                ev.record()
                self._events[b].append(ev)
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _adopt(self, i):
        """make parameter i's gradient the view into its bucket (a no-op when the kernel wrote it there)"""
        p, v = self.params[i], self._views[i]
        g = p.grad
        if g is None:
            v.zero_()
        elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
            v.copy_(g)
            self._copied.append(i)
        p.grad = v

    def _launch(self, b):
        for ev in self._events[b]:
            torch.cuda.current_stream().wait_event(ev)
        self._work[b] = dist.all_reduce(self._flat[b], op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM,
                                        group=self.group, async_op=True)

    def finish(self):
        """Call after backward(), before optimizer.step()."""
        if not self.active:
            return
        for b, idxs in enumerate(self.buckets):
            if self._work[b] is None:   # no hooks (overlap off) or a parameter that received no gradient
                for i in idxs:
                    self._adopt(i)
                self._launch(b)
        for b in range(len(self.buckets)):
            self._work[b].wait()
            if not self._avg and self.world > 1:
                self._flat[b].mul_(1.0 / self.world)
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            F_.GRAD_VIEWS.pop(id(p), None)


def all_reduce_mean_scalars(values, device=None, process_group=None):
    """Mean over ranks of a few Python floats (validation losses, logging scalars): one tiny collective."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return [float(v) for v in values]
    dev = device if (device is not None and dist.get_backend(process_group) == "nccl") else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    t /= dist.get_world_size(process_group)
    return [float(v) for v in t.tolist()]


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)
    sp = getattr(module, "_step_prep", None)   # packed weights derived before the broadcast are stale now
    if sp is not None:
        sp.invalidate()


def shard_batch(n_items: int, rank: int, world: int):
    """Contiguous, near-equal shard [lo, hi) of a global batch."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
