"""torch.autograd.Function wrappers over the C ABI of liblic_hip.so.

PyTorch here is plumbing only: it owns device memory, streams and the autograd graph.  Every
arithmetic step of the hot path runs in the hand-written gfx950 kernels behind `_lib`.
Tensors cross this boundary as NCHW-*logical* torch tensors whose physical layout is NHWC
(channels_last); `_nhwc()` is a zero-copy permute when that already holds.

No CPU fallback: a non-CUDA tensor or a missing extension raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib as L

PEDESTAL = float(2.0 ** -36)  # compressai reparam_offset ** 2
LIKELIHOOD_BOUND = 1e-9

# bench.py sets this to a list to collect (kernel, flops, activation_bytes, start_evt, end_evt) for
# every MFMA launch; HIP events are recorded on the stream the kernels are launched on.
PROFILE = None           # list -> every MFMA launch appends (kernel name, FLOP, bytes, start event, end event)
PROFILE_MIN_FLOP = 0.0   # launches below this many algorithmic FLOP are not bracketed (event overhead)
# Launch-plan overrides written into every descriptor (lic.h: force_bm / force_tn / force_split and
# force_tm / force_tn / force_split; None = automatic).  The automatic tile depends on the batch, so the
# parity tests use these to run every variant the full-size workloads dispatch against the oracle at small
# sizes; codec.py pins one variant so encoder and decoder build identical tables.
FORCE_IGEMM = None       # (bm, tn, split[, bf16 DMA ring]) -- 0 entries stay automatic
FORCE_WGRAD = None       # (tm, tn, split)
KERNEL_TRACE = None      # set -> the rocprofv3 name of every MFMA kernel variant launched is added
# id(parameter) -> (weakref, persistent gradient slot inside a data-parallel all-reduce bucket)
# (parallel.GradientAllReducer); the weight-gradient kernels then write straight into the bucket
GRAD_VIEWS = {}
# (id(parameter), kind) -> (parameter version, derived buffer, weakref): packed weights, GDN re-parametrisations
# ... refreshed once per optimizer step by prep.StepPrep (one lic_prep_run launch); see `prepared`
PREPARED = {}
# a second stream the layers may put work on that nothing downstream waits for soon (models.py sets it around the encoder
# when the model overlaps its branches; None otherwise): functional_bf16's early stem columns
AUX_STREAM = None


# ------------------------------------------------------------------------------------------
# plumbing helpers
# ------------------------------------------------------------------------------------------
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream (the raw-handle query is ~20x cheaper than building a
    torch.cuda.Stream object: ~400 launches per step)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------
# reductions left pending until the end of a backward pass (lic_reduce_batch, include/lic.h)
# ------------------------------------------------------------------------------------------
DEFER_REDUCTIONS = os.environ.get("LIC_DEFER_REDUCTIONS", "1") != "0"
# ... of the fp32 path too?  Off: its slab reductions are bandwidth-sized (60-240 MB each at config 2) and ran beside
# the other stream's kernels; batched at the end of backward they sit on the critical path (1412 vs 1431 img/s, same box)
DEFER_FP32 = os.environ.get("LIC_DEFER_FP32", "0") == "1"
_PENDING_JOBS = []   # L.ReduceJob of the running backward pass
_PENDING_KEEP = []   # tensors they name: partial sums, outputs, parameters
_PENDING_SEEN = set()


_PENDING_LATE = []   # tensors of an EARLY flush on another stream: released at the end of the pass
# flush_point active?  Only while plan.StepPlan records a step (it sets this): the replayed step gains ~1 % (9771 / 9772 vs
# 9729 / 9642 img/s at config 3, same box); for the eager, host-paced step no gain could be measured (its run-to-run
# spread on a shared host is larger than the effect), so it keeps the single flush at the end of the pass
EARLY_FLUSH = os.environ.get("LIC_EARLY_FLUSH", "0") == "1"
# True while plan.StepPlan runs the step it records (and its warm-up steps): stream placements that pay replayed but not
# host-paced (models.py: the factorised likelihood on the second stream)
PLAN_RECORDING = False


def flush_reductions(early_on=None):
    """launch every pending reduction (one lic_reduce_batch).  Autograd calls this at the end of a backward pass, on the
    caller's stream, after that stream has been made to wait for every stream gradients were produced on.
    `early_on` (a stream; flush_point's backward): launch what is pending so far on THAT stream, after everything queued
    on the current one -- the batched reduction holds no LDS and few registers, so unlike the weight-gradient launches it
    does run beside the data-gradient chain that continues on the current stream.  The tensors it reads were allocated on
    their producers' streams: they are kept until the end of the pass, where the engine orders the caller's stream behind
    every stream of the pass."""
    if _PENDING_JOBS:
        n = len(_PENDING_JOBS)
        arr = (L.ReduceJob * n)(*_PENDING_JOBS)
        del _PENDING_JOBS[:]
        try:
            if early_on is not None:
                early_on.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(early_on):
                    L.check(L.load().lic_reduce_batch(arr, n, _stream()), "lic_reduce_batch")
                _PENDING_LATE.extend(_PENDING_KEEP)
            else:
                L.check(L.load().lic_reduce_batch(arr, n, _stream()), "lic_reduce_batch")
        finally:
            del _PENDING_KEEP[:]
    if early_on is None:
        del _PENDING_LATE[:]
        _PENDING_SEEN.clear()


class _FlushPointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, stream):
        ctx.stream = stream
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        if _PENDING_JOBS and ctx.stream is not None:
            flush_reductions(early_on=ctx.stream)
        return g, None


def flush_point(t: torch.Tensor, stream):
    """identity; in the backward pass, when the gradient of `t` is complete, the reductions pending so far (those of
    everything downstream of `t`) are launched on `stream` while the pass continues upstream of `t` on its own stream"""
    if not (EARLY_FLUSH and DEFER_REDUCTIONS and t.requires_grad and torch.is_grad_enabled() and stream is not None) or \
            GRAD_VIEWS:
        return t
    return _FlushPointFn.apply(t, stream)


def can_defer(*params) -> bool:
    """May the reductions behind the gradients of `params` wait for the end of this backward pass?  Only if nothing
    reads such a gradient earlier: no data-parallel bucket hooks (they fire per gradient), the parameter has no
    gradient yet (autograd would ADD to it on arrival) and has not been met before in this pass (two uses of one
    parameter are summed when the second arrives: everything pending is flushed first), and we are inside a backward
    pass of the autograd engine (the flush is its final callback).  The pending job names the gradient tensor's memory
    but holds no reference to the tensor: autograd adopts a returned gradient as `.grad` only while nobody else holds it
    (otherwise it would COPY it -- before the reduction has run)."""
    if not DEFER_REDUCTIONS or GRAD_VIEWS:
        return False
    ps = [p for p in params if p is not None]
    if not all(p.is_leaf for p in ps):   # a derived weight: its gradient is READ by the next backward node
        return False
    if any(p.grad is not None or id(p) in _PENDING_SEEN for p in ps):
        flush_reductions()
        return False
    try:
        torch.autograd.Variable._execution_engine.queue_callback(flush_reductions)
    except RuntimeError:
        return False
    _PENDING_SEEN.update(id(p) for p in ps)
    return True


def defer(job, *keep):
    _PENDING_JOBS.append(job)
    _PENDING_KEEP.extend(t for t in keep if t is not None)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.LicError("neural_image_compression_amd runs on MI355X only: got a non-CUDA tensor "
                             "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")
        if t is not None and t.device.index != torch.cuda.current_device():
            # kernels are launched on the CURRENT device's stream (_stream): a tensor of another GPU would be
            # read through the wrong device's queue
            raise L.LicError(f"tensor on cuda:{t.device.index} but the current device is cuda:"
                             f"{torch.cuda.current_device()}: call torch.cuda.set_device(...) first (one process per GPU)")
        if t is not None and t.dtype != torch.float32:
            raise L.LicError(f"fp32 tensors expected, got {t.dtype}")


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """NCHW-logical -> [B,H,W,C] contiguous (no copy if the tensor is channels_last already)."""
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw_view(t_nhwc: torch.Tensor) -> torch.Tensor:
    return t_nhwc.permute(0, 3, 1, 2)


def _permute3(src, dst, n, s, d):
    L.check(L.load().lic_permute3(_ptr(src), _ptr(dst), n[0], n[1], n[2], s[0], s[1], s[2], d[0], d[1],
                                  d[2], _stream()), "lic_permute3")


def _pack(src: torch.Tensor, taps: int, K: int, N: int, s_tap: int, s_k: int, s_n: int) -> torch.Tensor:
    """Weights -> the MFMA B-operand layout of lic_igemm: [tap][K/16][ceil32(N)][16], zero padded.
    Element (tap, k, n) is read from src[tap*s_tap + k*s_k + n*s_n]."""
    lib = L.load()
    out = torch.empty((lib.lic_packed_weight_floats(taps, K, N),), device=src.device, dtype=torch.float32)
    L.check(lib.lic_pack_weight(_ptr(src), _ptr(out), taps, K, N, s_tap, s_k, s_n, _stream()),
            "lic_pack_weight")
    return out


def _pack_dense(m: torch.Tensor) -> torch.Tensor:
    """Row-major [K][N] matrix -> packed B operand."""
    K, N = m.shape
    return _pack(m, 1, K, N, 0, N, 1)


def _pack_conv_weight(w: torch.Tensor, transposed_weight: bool, for_dgrad: bool) -> torch.Tensor:
    """`transposed_weight`: w is [Cin,Cout,kh,kw] (ConvTranspose2d), else [Cout,Cin,kh,kw].
    Forward contracts over the layer's input channels, dgrad over its output channels."""
    hit = prepared(w, "f32.dgrad" if for_dgrad else "f32.fwd")
    if hit is not None:
        return hit
    w = w.contiguous()
    d0, d1, kh, kw = w.shape
    taps = kh * kw
    # element (a, b, tap) of w at a*d1*taps + b*taps + tap
    if transposed_weight:
        cin, cout = d0, d1
        s_ci, s_co = d1 * taps, taps
    else:
        cout, cin = d0, d1
        s_co, s_ci = d1 * taps, taps
    if for_dgrad:
        return _pack(w, taps, cout, cin, 1, s_co, s_ci)
    return _pack(w, taps, cin, cout, 1, s_ci, s_co)


def _igemm(inp, w_packed, out, *, B, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, transposed,
           bias=None, prologue=0, epilogue=L.EPI_NONE, slope=0.01, tap_mask=0, out2=None, aux=None,
           aux2=None, aux3=None, res=None, in_ld=None, out_ld=None, out3=None):
    d = L.IgemmDesc()
    d.in_, d.w, d.bias, d.out, d.out2 = _ptr(inp), _ptr(w_packed), _ptr(bias), _ptr(out), _ptr(out2)
    d.aux, d.aux2, d.aux3, d.res = _ptr(aux), _ptr(aux2), _ptr(aux3), _ptr(res)
    d.in_ld = Cin if in_ld is None else in_ld
    d.out_ld = Cout if out_ld is None else out_ld
    d.out2_ld = d.aux_ld = d.aux2_ld = d.aux3_ld = d.res_ld = d.out3_ld = Cout
    d.out3 = _ptr(out3)
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, Hi, Wi, Cin, Ho, Wo, Cout
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.transposed, d.prologue, d.epilogue = int(transposed), prologue, epilogue
    d.tap_mask, d.slope = tap_mask, slope
    lib = L.load()
    ws = None
    fused = epilogue in (L.EPI_CONV_GDN, L.EPI_CONV_IGDN)
    if FORCE_IGEMM is not None:
        d.force_bm, d.force_tn, d.force_split = FORCE_IGEMM[:3]
    if KERNEL_TRACE is not None:
        KERNEL_TRACE.add(_kernel_name(lib.lic_igemm_kernel_name, d))
    if Ho * Wo <= 1024 and out2 is None and res is None:  # latent-side layers: allow split-K
        nbytes = lib.lic_igemm_workspace_bytes(C.byref(d))
        if nbytes:
            ws = torch.empty(nbytes // 4, device=inp.device, dtype=torch.float32)
            d.workspace, d.workspace_bytes = _ptr(ws), nbytes
    if PROFILE is None or 2.0 * B * Ho * Wo * Cout * Cin * kh * kw < PROFILE_MIN_FLOP:
        L.check(lib.lic_igemm(C.byref(d), _stream()), "lic_igemm")
        return
    bm, bn, macs = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    lib.lic_igemm_plan(C.byref(d), C.byref(bm), C.byref(bn), C.byref(macs))
    nm = C.create_string_buffer(96)
    lib.lic_igemm_kernel_name(C.byref(d), nm, 96)  # as rocprofv3 prints it
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.lic_igemm(C.byref(d), _stream()), "lic_igemm")
    e1.record()
    act_bytes = 4 * (B * Hi * Wi * Cin + B * Ho * Wo * Cout)
    flops = 2 * macs.value
    if fused:  # conv + the channel pool of the GDN that follows it
        flops += 2 * B * Ho * Wo * Cout * Cout
        act_bytes += 4 * 2 * B * Ho * Wo * Cout
    PROFILE.append((nm.value.decode(), flops, act_bytes, e0, e1))


def _kernel_name(fn, d) -> str:
    nm = C.create_string_buffer(96)
    L.check(fn(C.byref(d), nm, 96), "kernel_name")
    return nm.value.decode()


def _timed(name, flops, act_bytes, launch):
    """run `launch()`; bracket it with HIP events when bench.py's PROFILE list is active"""
    if PROFILE is None or flops < PROFILE_MIN_FLOP:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    PROFILE.append((name, flops, act_bytes, e0, e1))


def _wgrad(p, g, dst, *, B, Hs, Ws, Cp, Hl, Wl, Cg, kh, kw, stride, pad, g_is_row, dst_sm, dst_sn,
           dst_stap, sq_p=0, sq_g=0, scale=1.0, job=None):
    """`job` (an L.ReduceJob): launch the MFMA kernel only and leave the slab reduction pending (`defer`; the caller may
    still set the job's epilogue / index-map fields); None: reduce right away"""
    d = L.WgradDesc()
    d.p, d.g, d.dst = _ptr(p), _ptr(g), _ptr(dst)
    d.p_ld, d.g_ld = Cp, Cg
    d.dst_sm, d.dst_sn, d.dst_stap = dst_sm, dst_sn, dst_stap
    d.B, d.Hs, d.Ws, d.Cp, d.Hl, d.Wl, d.Cg = B, Hs, Ws, Cp, Hl, Wl, Cg
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.g_is_row, d.sq_p, d.sq_g, d.scale = int(g_is_row), sq_p, sq_g, scale
    lib = L.load()
    if FORCE_WGRAD is not None:
        d.force_tm, d.force_tn, d.force_split = FORCE_WGRAD
    if KERNEL_TRACE is not None:
        KERNEL_TRACE.add(_kernel_name(lib.lic_wgrad_kernel_name, d))
    nbytes = lib.lic_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty((max(nbytes, 4) + 3) // 4, device=p.device, dtype=torch.float32)
    timed = not (PROFILE is None or 2.0 * B * Hs * Ws * kh * kw * Cp * Cg < PROFILE_MIN_FLOP)
    if job is not None:
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(lib.lic_wgrad_partial(C.byref(d), _ptr(ws), nbytes, C.byref(job), _stream()), "lic_wgrad_partial")
        defer(job, ws, p, g)   # (not `dst`: autograd adopts a gradient tensor only if nobody else holds it)
        if timed:
            e1.record()
            nm = C.create_string_buffer(96)
            lib.lic_wgrad_kernel_name(C.byref(d), nm, 96)
            PROFILE.append((nm.value.decode(), 2 * B * Hs * Ws * kh * kw * Cp * Cg,
                            4 * (B * Hs * Ws * Cp + B * Hl * Wl * Cg), e0, e1))
        return
    if not timed:
        L.check(lib.lic_wgrad(C.byref(d), _ptr(ws), nbytes, _stream()), "lic_wgrad")
        return
    nm = C.create_string_buffer(96)
    lib.lic_wgrad_kernel_name(C.byref(d), nm, 96)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()  # the MFMA kernel alone; the slab reduction runs after the second event
    L.check(lib.lic_wgrad_stage(C.byref(d), _ptr(ws), nbytes, 1, _stream()), "lic_wgrad_stage")
    e1.record()
    L.check(lib.lic_wgrad_stage(C.byref(d), _ptr(ws), nbytes, 2, _stream()), "lic_wgrad_stage")
    flops = 2 * B * Hs * Ws * kh * kw * Cp * Cg
    PROFILE.append((nm.value.decode(), flops, 4 * (B * Hs * Ws * Cp + B * Hl * Wl * Cg), e0, e1))


def _colsum(t2d: torch.Tensor, P: int, Cc: int, scale: float = 1.0, deferred: bool = False, reparam=None) -> torch.Tensor:
    """column sums; `deferred`: the second stage joins the backward pass's batched reduction (`defer`), optionally
    followed by the GDN re-parametrisation's backward (`reparam` = (parameter, bound))"""
    lib = L.load()
    nbytes = lib.lic_colsum_workspace_bytes(P, Cc)
    ws = torch.empty((nbytes + 3) // 4, device=t2d.device, dtype=torch.float32)
    out = torch.empty((Cc,), device=t2d.device, dtype=torch.float32)
    if deferred:
        job = L.ReduceJob()
        L.check(lib.lic_colsum_partial(_ptr(t2d), Cc, P, Cc, scale, _ptr(out), _ptr(ws), nbytes, C.byref(job), _stream()),
                "lic_colsum_partial")
        if reparam is not None:
            job.epilogue, job.param, job.bound = L.REDUCE_EPI_REPARAM, reparam[0].data_ptr(), reparam[1]
        defer(job, ws, t2d, reparam[0] if reparam is not None else None)
        return out
    L.check(lib.lic_colsum(_ptr(t2d), Cc, P, Cc, scale, _ptr(out), _ptr(ws), nbytes, _stream()),
            "lic_colsum")
    return out


def prepared(param: torch.Tensor, kind: str):
    """The buffer of `kind` that prep.StepPrep derived from `param`, if it was built from the parameter's
    current version (optimizer.step() bumps the version; autograd hands backward the same Parameter object
    it saw in forward); None -> the caller derives it on the fly."""
    e = PREPARED.get((id(param), kind))
    if e is not None and e[0] == param._version and e[2]() is param:
        return e[1]
    return None


def grad_like(param: torch.Tensor) -> torch.Tensor:
    """Destination for `param`'s gradient: its slot in the all-reduce bucket when data parallelism registered one,
    no gradient has been accumulated yet and the slot has not been handed out in this step (autograd then adopts
    the slot as `.grad` without a copy), else a fresh contiguous tensor.  The slot goes out AT MOST ONCE per step
    (entry[2], cleared by GradientAllReducer.reset()): a parameter used by two nodes of one graph -- a layer called
    twice, shared weights -- would otherwise get the same destination for both weight-gradient launches, the second
    overwriting the first, and autograd would add the two aliased handles (2x the last contribution, silently)."""
    e = GRAD_VIEWS.get(id(param)) if GRAD_VIEWS else None
    if e is not None and e[0]() is param and param.grad is None and not e[2]:
        e[2] = True
        return e[1].detach()   # a fresh handle on the slot: autograd adopts a gradient only if nobody else holds it
    return torch.empty_like(param, memory_format=torch.contiguous_format)


def _reparam_bwd2(beta_c, dbe, beta_bound, gamma_c, dge, gamma_bound):
    """gradients of beta / gamma through compressai's NonNegativeParametrizer from the gradients
    of beta_eff / gamma_eff: both parameters of a GDN in one launch"""
    dbeta = torch.empty_like(beta_c) if beta_c is not None else None
    dgamma = torch.empty_like(gamma_c) if gamma_c is not None else None
    if dbeta is None and dgamma is None:
        return None, None
    L.check(L.load().lic_gdn_reparam_bwd2(_ptr(beta_c), _ptr(dbe), _ptr(dbeta), 0 if dbeta is None else dbeta.numel(),
                                          beta_bound, _ptr(gamma_c), _ptr(dge), _ptr(dgamma),
                                          0 if dgamma is None else dgamma.numel(), gamma_bound, _stream()),
            "lic_gdn_reparam_bwd2")
    return dbeta, dgamma


def _leaky_bwd(y, dy, slope):
    dx = torch.empty_like(y)
    L.check(L.load().lic_leaky_bwd(_ptr(y), _ptr(dy), _ptr(dx), y.numel(), slope, _stream()),
            "lic_leaky_bwd")
    return dx


def conv_out_size(H, W, k, stride, pad, transposed, out_pad=0):
    if transposed:
        return (H - 1) * stride - 2 * pad + k + out_pad, (W - 1) * stride - 2 * pad + k + out_pad
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


# ------------------------------------------------------------------------------------------
# convolution / transposed convolution with fused bias (+ LeakyReLU)
# ------------------------------------------------------------------------------------------
class _ConvFn(torch.autograd.Function):
    """nn.Conv2d / nn.ConvTranspose2d (+ optional fused LeakyReLU) for channel counts >= 4.
    Reference call sites: Components.py:10-16,39-45,69-73,99-103; Layers.py:21,38-43,74-76,99-103;
    ParametersModels.py:22-34; ContextModels.py:19-20."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad, transposed, leaky, slope, tap_mask, res, out_view=None):
        _require_cuda(x, weight, bias, res)
        xh = _nhwc(x)
        B, Hi, Wi, Cin = xh.shape
        kh, kw = weight.shape[2], weight.shape[3]
        Cout = weight.shape[1] if transposed else weight.shape[0]
        Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, transposed, out_pad)
        wp = _pack_conv_weight(weight, transposed, for_dgrad=False)
        out_ld = None
        if out_view is not None:
            # the caller's channel range of a wider NHWC buffer (two producers fill one tensor: no torch.cat)
            if leaky or res is not None or tuple(out_view.shape) != (B, Ho, Wo, Cout) or out_view.stride(3) != 1 or \
                    out_view.stride(1) != Wo * out_view.stride(2) or out_view.stride(0) != Ho * out_view.stride(1):
                raise ValueError("out_view must be a [B,Ho,Wo,Cout] channel slice of a contiguous NHWC buffer")
            out, out_ld = out_view, out_view.stride(2)
        else:
            out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        resh = None if res is None else _nhwc(res)
        # leaky + residual: `out` keeps leaky(conv) for the backward mask, `out2` = out + res
        out2 = torch.empty_like(out) if (leaky and res is not None) else None
        _igemm(xh, wp, out, B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, Cout=Cout, kh=kh, kw=kw,
               stride=stride, pad=pad, transposed=transposed, bias=bias,
               epilogue=L.EPI_LEAKY if leaky else L.EPI_NONE, slope=slope, tap_mask=tap_mask,
               res=resh, out2=out2, out_ld=out_ld)
        ctx.save_for_backward(xh, weight, out if leaky else None)
        ctx.cfg = (stride, pad, transposed, leaky, slope, tap_mask, bias is not None, res is not None)
        return _nchw_view(out2 if out2 is not None else out)

    @staticmethod
    def backward(ctx, gy):
        xh, weight, yh = ctx.saved_tensors
        stride, pad, transposed, leaky, slope, tap_mask, has_bias, has_res = ctx.cfg
        g = _nhwc(gy)
        if leaky:
            g = _leaky_bwd(yh, g, slope)
        elif hasattr(gy, "_lic_colsum_partial") and not has_res:
            g._lic_colsum_partial = gy._lic_colsum_partial
        dx, dw, db = _conv_backward(xh, weight, g, stride, pad, transposed, tap_mask,
                                    ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                    has_bias and ctx.needs_input_grad[2])
        return dx, dw, db, None, None, None, None, None, None, None, (gy if has_res else None), None


def _conv_backward(xh, weight, g, stride, pad, transposed, tap_mask, need_dx, need_dw, need_db):
    """Input / weight / bias gradients of a conv (or transposed conv) from the NHWC output gradient."""
    B, Hi, Wi, Cin = xh.shape
    _, Ho, Wo, Cout = g.shape
    kh, kw = weight.shape[2], weight.shape[3]
    dx = dw = db = None
    if need_dx:
        wp = _pack_conv_weight(weight, transposed, for_dgrad=True)
        dxh = torch.empty_like(xh)
        # the data gradient of a conv is the transposed gather and vice versa
        _igemm(g, wp, dxh, B=B, Hi=Ho, Wi=Wo, Cin=Cout, Ho=Hi, Wo=Wi, Cout=Cin, kh=kh, kw=kw,
               stride=stride, pad=pad, transposed=not transposed, tap_mask=tap_mask)
        dx = _nchw_view(dxh)
    # the slab reduction of the weight gradient and the second stage of the bias sum wait for the end of the backward
    # pass when nothing can read these gradients earlier (can_defer): one batched launch for the whole pass
    dfr = DEFER_FP32 and (need_dw or need_db) and can_defer(weight)
    if need_dw:
        dw = grad_like(weight)
        taps = kh * kw
        job = L.ReduceJob() if dfr else None
        if transposed:  # weight [Cin,Cout,kh,kw]; small grid = input, gathered = grad
            _wgrad(xh, g, dw, B=B, Hs=Hi, Ws=Wi, Cp=Cin, Hl=Ho, Wl=Wo, Cg=Cout, kh=kh, kw=kw,
                   stride=stride, pad=pad, g_is_row=False, dst_sm=Cout * taps, dst_sn=taps, dst_stap=1, job=job)
        else:  # weight [Cout,Cin,kh,kw]; small grid = output grad, gathered = input
            _wgrad(g, xh, dw, B=B, Hs=Ho, Ws=Wo, Cp=Cout, Hl=Hi, Wl=Wi, Cg=Cin, kh=kh, kw=kw,
                   stride=stride, pad=pad, g_is_row=True, dst_sm=taps, dst_sn=Cin * taps, dst_stap=1, job=job)
    if need_db:
        db = _bias_grad(g, B * Ho * Wo, Cout, dfr)
    return dx, dw, db


def _bias_grad(g, P, Cout, deferred=False):
    """column sums of the output gradient; from the GDN backward's per-workgroup partials when it left them"""
    part = getattr(g, "_lic_colsum_partial", None)
    if part is not None and part.shape[1] == Cout:
        return _colsum(part, part.shape[0], Cout, deferred=deferred)
    return _colsum(g, P, Cout, deferred=deferred)


def conv2d(x, weight, bias, stride=1, padding=0, leaky=False, slope=0.01, tap_mask=0, residual=None, out=None):
    """`out`: optional [B,Ho,Wo,Cout] channel slice of a wider contiguous NHWC buffer to write into (the kernel's
    row pitch `out_ld`); the result is then a view of that buffer."""
    return _ConvFn.apply(x, weight, bias, stride, padding, 0, False, leaky, slope, tap_mask, residual, out)


class _JoinChannelsFn(torch.autograd.Function):
    """`torch.cat([a, b], dim=1)` (Models.py:73) when a and b were WRITTEN as the two channel ranges of `buf`
    (conv2d(..., out=slice)): the forward is a view, the backward hands each producer its slice of the gradient."""

    @staticmethod
    def forward(ctx, a, b, buf):
        ca, cb = a.shape[1], b.shape[1]
        if buf.shape[-1] != ca + cb or a.data_ptr() != buf.data_ptr() or \
                b.data_ptr() != buf.data_ptr() + buf.element_size() * ca:
            raise ValueError("join_channels: the operands are not the two channel ranges of the buffer")
        ctx.ca = ca
        return _nchw_view(buf)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.ca], g[:, ctx.ca:], None


def join_channels(a, b, buf):
    return _JoinChannelsFn.apply(a, b, buf)


def pack_conv_weight(weight: torch.Tensor) -> torch.Tensor:
    """The forward B-operand layout of an nn.Conv2d weight, for `conv2d_prepacked`."""
    _require_cuda(weight)
    return _pack_conv_weight(weight.detach(), False, for_dgrad=False)


@torch.no_grad()
def conv2d_prepacked(x, w_packed, bias, cout, kernel, stride=1, padding=0, leaky=False, slope=0.01, tap_mask=0,
                     pin_tile=False):
    """Inference-only conv2d whose weight was packed once with `pack_conv_weight` (no autograd, no per-call
    packing): same kernel, same operands, hence the same bits as `conv2d`.  The serial context decoder calls
    four layers h*w times with unchanged weights (codec.ContextCodec)."""
    _require_cuda(x, w_packed, bias)
    xh = _nhwc(x)
    B, Hi, Wi, Cin = xh.shape
    Ho, Wo = conv_out_size(Hi, Wi, kernel, stride, padding, False)
    out = torch.empty((B, Ho, Wo, cout), device=x.device, dtype=torch.float32)
    global FORCE_IGEMM
    saved = FORCE_IGEMM
    if pin_tile and Cin % 4 == 0 and cout % 64 == 0:
        # one tile variant whatever the batch (the entropy coder's encoder sees all pixels at once, its decoder a
        # wavefront of them).  Every variant accumulates an output element over K in the same order, so they
        # agree bit for bit by construction; pinning just stops the agreement from resting on that argument.
        FORCE_IGEMM = (64, 1, 0)
    try:
        _igemm(xh, w_packed, out, B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, Cout=cout, kh=kernel, kw=kernel,
               stride=stride, pad=padding, transposed=False, bias=bias,
               epilogue=L.EPI_LEAKY if leaky else L.EPI_NONE, slope=slope, tap_mask=tap_mask)
    finally:
        FORCE_IGEMM = saved
    return _nchw_view(out)


def conv_transpose2d(x, weight, bias, stride=1, padding=0, output_padding=0, leaky=False, slope=0.01,
                     residual=None):
    return _ConvFn.apply(x, weight, bias, stride, padding, output_padding, True, leaky, slope, 0, residual)


# ------------------------------------------------------------------------------------------
# image-side layers (3 channels): patches <-> columns + one dense GEMM
# ------------------------------------------------------------------------------------------
def _kpad(kh, kw, c):
    return (kh * kw * c + 3) // 4 * 4


def _image_conv_columns(xh, weight, stride, pad):
    """im2col of the few-channel input + the packed [Kp][Cout] weight matrix; returns (col, wp_packed, Ho, Wo, Kp)."""
    B, Hi, Wi, Cin = xh.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, False)
    Kp = _kpad(kh, kw, Cin)
    P = B * Ho * Wo
    col = torch.empty((P, Kp), device=xh.device, dtype=torch.float32)
    L.check(L.load().lic_im2col(_ptr(xh), _ptr(col), B, Hi, Wi, Cin, Ho, Wo, kh, kw, stride, pad, Kp,
                                _stream()), "lic_im2col")
    taps = kh * kw
    wpk = prepared(weight, "f32.stem")
    if wpk is None:
        wp = torch.zeros((Kp, Cout), device=xh.device, dtype=torch.float32)
        # wp[tap*Cin + c][co] = w[co][c][tap]
        _permute3(weight.contiguous(), wp, (taps, Cin, Cout), (1, taps, Cin * taps), (Cin * Cout, Cout, 1))
        wpk = _pack_dense(wp)
    return col, wpk, Ho, Wo, Kp


def _image_conv_backward(col, weight, g, stride, pad, in_shape, need_dx, need_dw, need_db):
    B, Hi, Wi, Cin = in_shape
    _, Ho, Wo, Cout = g.shape
    _, _, kh, kw = weight.shape
    taps, Kp, P = kh * kw, col.shape[1], B * Ho * Wo
    lib = L.load()
    dx = dw = db = None
    if need_dx:
        wpT = torch.zeros((Cout, Kp), device=g.device, dtype=torch.float32)
        _permute3(weight.contiguous(), wpT, (Cout, Cin, taps), (Cin * taps, taps, 1), (Kp, 1, Cin))
        dcol = torch.empty((P, Kp), device=g.device, dtype=torch.float32)
        _igemm(g, _pack_dense(wpT), dcol, B=1, Hi=1, Wi=P, Cin=Cout, Ho=1, Wo=P, Cout=Kp, kh=1, kw=1, stride=1,
               pad=0, transposed=False)
        dxh = torch.empty((B, Hi, Wi, Cin), device=g.device, dtype=torch.float32)
        L.check(lib.lic_col2im(_ptr(dcol), None, _ptr(dxh), B, Ho, Wo, Cin, Hi, Wi, kh, kw, stride,
                               pad, Kp, _stream()), "lic_col2im")
        dx = _nchw_view(dxh)
    dfr = DEFER_FP32 and (need_dw or need_db) and can_defer(weight)
    if need_dw:
        dw = grad_like(weight)
        if dfr:
            # the pending reduction writes dw[co][c][tap] itself: row m = tap * Cin + c of the [Kp][Cout] product goes to
            # tap + c * taps, column co to co * Cin * taps (what the permute launch below does otherwise)
            job = L.ReduceJob()
            _wgrad(col, g, dw, B=1, Hs=1, Ws=P, Cp=Kp, Hl=1, Wl=P, Cg=Cout, kh=1, kw=1, stride=1, pad=0,
                   g_is_row=False, dst_sm=Cout, dst_sn=1, dst_stap=0, job=job)
            job.mdiv, job.sm, job.smr, job.sn, job.Mvalid = Cin, 1, taps, Cin * taps, taps * Cin
        else:
            tmp = torch.empty((Kp, Cout), device=g.device, dtype=torch.float32)
            _wgrad(col, g, tmp, B=1, Hs=1, Ws=P, Cp=Kp, Hl=1, Wl=P, Cg=Cout, kh=1, kw=1, stride=1, pad=0,
                   g_is_row=False, dst_sm=Cout, dst_sn=1, dst_stap=0)
            _permute3(tmp, dw, (taps, Cin, Cout), (Cin * Cout, Cout, 1), (1, taps, Cin * taps))
    if need_db:
        db = _bias_grad(g, P, Cout, dfr)
    return dx, dw, db


class _ImageConvFn(torch.autograd.Function):
    """nn.Conv2d whose INPUT has few channels (the RGB stem: Components.py:10; Layers.py:38,43
    inside Encoder3x3's first block).  im2col (HBM-bound) + MFMA GEMM with K = kh*kw*C."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, leaky, slope):
        _require_cuda(x, weight, bias)
        xh = _nhwc(x)
        B = xh.shape[0]
        Cout = weight.shape[0]
        col, wp, Ho, Wo, Kp = _image_conv_columns(xh, weight, stride, pad)
        P = B * Ho * Wo
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        _igemm(col, wp, out, B=1, Hi=1, Wi=P, Cin=Kp, Ho=1, Wo=P, Cout=Cout, kh=1, kw=1, stride=1,
               pad=0, transposed=False, bias=bias, epilogue=L.EPI_LEAKY if leaky else L.EPI_NONE,
               slope=slope)
        ctx.save_for_backward(col, weight, out if leaky else None)
        ctx.cfg = (stride, pad, leaky, slope, tuple(xh.shape), bias is not None)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, gy):
        col, weight, yh = ctx.saved_tensors
        stride, pad, leaky, slope, in_shape, has_bias = ctx.cfg
        g = _nhwc(gy)
        if leaky:
            g = _leaky_bwd(yh, g, slope)
        elif hasattr(gy, "_lic_colsum_partial"):
            g._lic_colsum_partial = gy._lic_colsum_partial
        dx, dw, db = _image_conv_backward(col, weight, g, stride, pad, in_shape, ctx.needs_input_grad[0],
                                          ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2])
        return dx, dw, db, None, None, None, None


class _ImageConvGDNFn(torch.autograd.Function):
    """The RGB stem and the GDN behind it (Components.py:10-11) with the pool fused into the GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, beta, gamma, stride, pad, inverse, beta_bound, gamma_bound, pedestal, keep=True):
        _require_cuda(x, weight, bias, beta, gamma)
        xh = _nhwc(x)
        B = xh.shape[0]
        Cout = weight.shape[0]
        col, wp, Ho, Wo, Kp = _image_conv_columns(xh, weight, stride, pad)
        P = B * Ho * Wo
        beta_e, gT = _gdn_operands(beta, gamma, beta_bound, gamma_bound, pedestal)
        y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        conv_out = torch.empty_like(y) if keep else None   # (only the backward pass reads these two)
        norm = torch.empty_like(y) if keep else None
        _igemm(col, wp, y, B=1, Hi=1, Wi=P, Cin=Kp, Ho=1, Wo=P, Cout=Cout, kh=1, kw=1, stride=1,
               pad=0, transposed=False, bias=bias, epilogue=L.EPI_CONV_IGDN if inverse else L.EPI_CONV_GDN,
               out2=norm, out3=conv_out, aux=gT, aux2=beta_e)
        ctx.save_for_backward(col, weight, conv_out, norm, beta, gamma)
        ctx.cfg = (stride, pad, tuple(xh.shape), inverse, beta_bound, gamma_bound, bias is not None, pedestal)
        return _nchw_view(y)

    @staticmethod
    def backward(ctx, gy):
        col, weight, conv_out, norm, beta, gamma = ctx.saved_tensors
        stride, pad, in_shape, inverse, beta_bound, gamma_bound, has_bias, pedestal = ctx.cfg
        need = ctx.needs_input_grad
        g_conv, dbeta, dgamma = _gdn_backward(conv_out, norm, beta, gamma, _nhwc(gy), inverse,
                                              beta_bound, gamma_bound, pedestal, True, need[3], need[4])
        dx, dw, db = _image_conv_backward(col, weight, g_conv, stride, pad, in_shape, need[0], need[1],
                                          has_bias and need[2])
        return dx, dw, db, dbeta, dgamma, None, None, None, None, None, None, None


class _ImageConvTFn(torch.autograd.Function):
    """nn.ConvTranspose2d whose OUTPUT has few channels (the RGB head: Components.py:45;
    Components.py:60 via Layers.py:21).  Dense MFMA GEMM to per-tap columns + col2im gather."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad):
        _require_cuda(x, weight, bias)
        xh = _nhwc(x)
        B, Hi, Wi, Cin = xh.shape
        _, Cout, kh, kw = weight.shape
        Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, True, out_pad)
        taps, Kp, P = kh * kw, _kpad(kh, kw, Cout), B * Hi * Wi
        lib = L.load()
        wpk = prepared(weight, "f32.head")
        if wpk is None:
            wp = torch.zeros((Cin, Kp), device=x.device, dtype=torch.float32)
            # wp[ci][tap*Cout + co] = w[ci][co][tap]
            _permute3(weight.contiguous(), wp, (Cin, Cout, taps), (Cout * taps, taps, 1), (Kp, 1, Cout))
            wpk = _pack_dense(wp)
        col = torch.empty((P, Kp), device=x.device, dtype=torch.float32)
        _igemm(xh, wpk, col, B=1, Hi=1, Wi=P, Cin=Cin, Ho=1, Wo=P, Cout=Kp, kh=1, kw=1, stride=1, pad=0,
               transposed=False)
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        L.check(lib.lic_col2im(_ptr(col), _ptr(bias), _ptr(out), B, Hi, Wi, Cout, Ho, Wo, kh, kw, stride,
                               pad, Kp, _stream()), "lic_col2im")
        ctx.save_for_backward(xh, weight)
        ctx.cfg = (stride, pad, (Ho, Wo), bias is not None)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, gy):
        xh, weight = ctx.saved_tensors
        stride, pad, (Ho, Wo), has_bias = ctx.cfg
        g = _nhwc(gy)
        B, Hi, Wi, Cin = xh.shape
        _, Cout, kh, kw = weight.shape
        taps, Kp, P = kh * kw, _kpad(kh, kw, Cout), B * Hi * Wi
        lib = L.load()
        dcol = torch.empty((P, Kp), device=g.device, dtype=torch.float32)
        L.check(lib.lic_im2col(_ptr(g), _ptr(dcol), B, Ho, Wo, Cout, Hi, Wi, kh, kw, stride, pad, Kp,
                               _stream()), "lic_im2col")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wpk = prepared(weight, "f32.head_dx")
            if wpk is None:
                wpT = torch.zeros((Kp, Cin), device=g.device, dtype=torch.float32)
                _permute3(weight.contiguous(), wpT, (taps, Cout, Cin), (1, taps, Cout * taps),
                          (Cout * Cin, Cin, 1))
                wpk = _pack_dense(wpT)
            dxh = torch.empty_like(xh)
            _igemm(dcol, wpk, dxh, B=1, Hi=1, Wi=P, Cin=Kp, Ho=1, Wo=P, Cout=Cin, kh=1, kw=1, stride=1,
                   pad=0, transposed=False)
            dx = _nchw_view(dxh)
        dfr = DEFER_FP32 and (ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])) and can_defer(weight)
        if ctx.needs_input_grad[1]:
            dw = grad_like(weight)
            # rows = the 80 columns of dcol, cols = the Cin channels of x: ONE 128x192 tile spans the whole product,
            # so each operand is streamed once per split (x on the rows needed 3 x 2 tiles of 64x64: 378 -> ~200 us)
            if dfr:
                # pending reduction with the column -> (tap, colour) map: row m = tap * Cout + c of the [Kp][Cin] product
                # (g_is_row: rows = dcol's columns) goes to tap + c * taps of dw[ci][c][tap], column ci to ci * Cout * taps
                job = L.ReduceJob()
                _wgrad(xh, dcol, dw, B=1, Hs=1, Ws=P, Cp=Cin, Hl=1, Wl=P, Cg=Kp, kh=1, kw=1, stride=1, pad=0,
                       g_is_row=True, dst_sm=1, dst_sn=Kp, dst_stap=0, job=job)
                job.mdiv, job.sm, job.smr, job.sn, job.Mvalid = Cout, 1, taps, Cout * taps, taps * Cout
            else:
                tmp = torch.empty((Cin, Kp), device=g.device, dtype=torch.float32)
                _wgrad(xh, dcol, tmp, B=1, Hs=1, Ws=P, Cp=Cin, Hl=1, Wl=P, Cg=Kp, kh=1, kw=1, stride=1, pad=0,
                       g_is_row=True, dst_sm=1, dst_sn=Kp, dst_stap=0)
                _permute3(tmp, dw, (Cin, taps, Cout), (Kp, Cout, 1), (Cout * taps, 1, taps))
        if has_bias and ctx.needs_input_grad[2]:
            db = _colsum(g, B * Ho * Wo, Cout, deferred=dfr)
        return dx, dw, db, None, None, None


def image_conv2d(x, weight, bias, stride, padding, leaky=False, slope=0.01):
    return _ImageConvFn.apply(x, weight, bias, stride, padding, leaky, slope)


def image_conv_transpose2d(x, weight, bias, stride, padding, output_padding):
    return _ImageConvTFn.apply(x, weight, bias, stride, padding, output_padding)


# ------------------------------------------------------------------------------------------
# standalone LeakyReLU (modules that are used outside the fused stacks)
# ------------------------------------------------------------------------------------------
class _LeakyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        _require_cuda(x)
        xc = x.contiguous()
        # y = leaky(x) via the select kernel: y = x > 0 ? x : slope * x
        y = _leaky_bwd(xc, xc, slope)
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return _leaky_bwd(y, gy.contiguous(), ctx.slope), None


def leaky_relu(x, slope=0.01):
    return _LeakyFn.apply(x, slope)


# ------------------------------------------------------------------------------------------
# GDN / IGDN (third-party compressai definition)
# ------------------------------------------------------------------------------------------
class _GDNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, beta, gamma, inverse, beta_bound, gamma_bound, pedestal, res, keep=True):
        _require_cuda(x, beta, gamma, res)
        lib = L.load()
        xh = _nhwc(x)
        B, H, W, Cc = xh.shape
        beta_e, gT = _gdn_operands(beta, gamma, beta_bound, gamma_bound, pedestal)
        out = torch.empty_like(xh)
        norm = torch.empty_like(xh) if keep else None   # (only the backward pass reads it)
        resh = None if res is None else _nhwc(res)
        P = B * H * W
        if lib.lic_gdn_supported(Cc):
            if KERNEL_TRACE is not None:
                KERNEL_TRACE.add(f"gdn_kernel<{Cc // 64}, {int(inverse)}>")
            _timed(f"gdn_kernel<{Cc // 64}, 0>", 2 * P * Cc * Cc, 4 * 3 * P * Cc,
                   lambda: L.check(lib.lic_gdn_fwd(_ptr(xh), _ptr(gT), _ptr(beta_e), _ptr(resh), _ptr(out), _ptr(norm),
                                                   P, Cc, int(inverse), _stream()), "lic_gdn_fwd"))
        else:
            _igemm(xh, gT, out, B=1, Hi=1, Wi=P, Cin=Cc, Ho=1, Wo=P, Cout=Cc, kh=1, kw=1, stride=1, pad=0,
                   transposed=False, bias=beta_e, prologue=1, epilogue=L.EPI_IGDN if inverse else L.EPI_GDN,
                   out2=norm, aux=xh, res=resh)
        ctx.save_for_backward(xh, norm, beta, gamma)
        ctx.cfg = (inverse, beta_bound, gamma_bound, res is not None, pedestal)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, gy):
        xh, norm, beta, gamma = ctx.saved_tensors
        inverse, beta_bound, gamma_bound, has_res, pedestal = ctx.cfg
        dxh, dbeta, dgamma = _gdn_backward(xh, norm, beta, gamma, _nhwc(gy), inverse, beta_bound,
                                           gamma_bound, pedestal, *ctx.needs_input_grad[:3])
        dres = gy if has_res else None
        dx = None if dxh is None else _nchw_view(dxh)
        if dx is not None and hasattr(dxh, "_lic_colsum_partial"):
            dx._lic_colsum_partial = dxh._lic_colsum_partial  # rides along to the producing conv's backward
        return dx, dbeta, dgamma, None, None, None, None, dres, None


def _gdn_operands(beta, gamma, beta_bound, gamma_bound, pedestal):
    """(beta_eff [C], packed gamma_eff^T: B operand [k = j][n = i] = gamma_eff[i][j]) of a GDN's forward pool:
    from the step preparation when it is current, else derived here."""
    beta_e, gT = prepared(beta, "f32.beta_e"), prepared(gamma, "f32.gdn_gT")
    if beta_e is None or gT is None:
        Cc = beta.numel()
        _, _, beta_e, gamma_e = _gdn_reparam(beta, gamma, beta_bound, gamma_bound, pedestal)
        gT = _pack(gamma_e, 1, Cc, Cc, 0, 1, Cc)
    return beta_e, gT


def _gdn_gamma_packed(gamma, gamma_bound, pedestal):
    """packed gamma_eff (row-major [j][i]): the backward contraction t . gamma_eff"""
    gp = prepared(gamma, "f32.gdn_g")
    if gp is None:
        lib = L.load()
        gamma_c = gamma.contiguous()
        gamma_e = torch.empty_like(gamma_c)
        L.check(lib.lic_gdn_reparam(_ptr(gamma_c), _ptr(gamma_e), gamma_c.numel(), gamma_bound, pedestal, _stream()),
                "lic_gdn_reparam")
        gp = _pack_dense(gamma_e)
    return gp


def _gdn_reparam(beta, gamma, beta_bound, gamma_bound, pedestal):
    """beta_eff [C], gamma_eff [C,C] of the non-negative re-parametrisation."""
    lib = L.load()
    Cc = beta.numel()
    beta_c, gamma_c = beta.contiguous(), gamma.contiguous()
    beta_e = torch.empty_like(beta_c)
    gamma_e = torch.empty_like(gamma_c)
    L.check(lib.lic_gdn_reparam(_ptr(beta_c), _ptr(beta_e), Cc, beta_bound, pedestal, _stream()),
            "lic_gdn_reparam")
    L.check(lib.lic_gdn_reparam(_ptr(gamma_c), _ptr(gamma_e), Cc * Cc, gamma_bound, pedestal,
                                _stream()), "lic_gdn_reparam")
    return beta_c, gamma_c, beta_e, gamma_e


def _gdn_backward(xh, norm, beta, gamma, g, inverse, beta_bound, gamma_bound, pedestal, need_dx, need_dbeta,
                  need_dgamma):
    """GDN / IGDN backward from the saved input x and pool `norm`; returns (dx NHWC, dbeta, dgamma)."""
    lib = L.load()
    B, H, W, Cc = xh.shape
    P = B * H * W
    t = torch.empty_like(xh)
    dxh = dbeta = dgamma = None
    cs_dx = None
    dbe = None
    beta_c, gamma_c = beta.contiguous(), gamma.contiguous()
    gp = _gdn_gamma_packed(gamma, gamma_bound, pedestal) if need_dx else None
    # the parameter-gradient reductions (and the re-parametrisation behind them) may wait for the end of the pass
    dfr = DEFER_FP32 and (need_dbeta or need_dgamma) and \
        can_defer(beta if need_dbeta else None, gamma if need_dgamma else None)
    dbeta_d = None
    if need_dx and lib.lic_gdn_supported(Cc):
        # the dedicated one-sweep kernel: t built from (g, x, norm) as the tile is loaded; it also emits
        # per-workgroup column sums of t (-> d beta) and of dx (-> the d bias of the conv in front)
        dxh = torch.empty_like(xh)
        rows = lib.lic_gdn_bwd_partial_rows(P)
        pt = torch.empty((rows, Cc), device=xh.device, dtype=torch.float32)
        pdx = torch.empty((rows, Cc), device=xh.device, dtype=torch.float32)
        if KERNEL_TRACE is not None:
            KERNEL_TRACE.add(f"gdn_bwd_kernel<{Cc // 64}, {int(inverse)}>")
        _timed(f"gdn_bwd_reg_kernel<{Cc // 64}>", 2 * P * Cc * Cc, 4 * 5 * P * Cc,
               lambda: L.check(lib.lic_gdn_bwd(_ptr(g), _ptr(xh), _ptr(norm), _ptr(gp), _ptr(dxh), _ptr(t), _ptr(pt),
                                               _ptr(pdx), P, Cc, int(inverse), _stream()), "lic_gdn_bwd"))
        if need_dbeta:
            if dfr:   # pending: second stage + re-parametrisation in the pass's batched reduction
                dbeta_d = _colsum(pt, rows, Cc, deferred=True, reparam=(beta_c, beta_bound)).view(beta_c.shape)
            else:
                dbe = _colsum(pt, rows, Cc)
        cs_dx = pdx
    elif need_dx and Cc % 4 == 0:
        # one launch: t = dL/dnorm built on the fly as the contraction's operand (and stored for the
        # parameter gradients), dx = g * rsqrt(norm) + 2 x (t . gamma) in the epilogue
        dxh = torch.empty_like(xh)
        _igemm(g, gp, dxh, B=1, Hi=1, Wi=P, Cin=Cc, Ho=1, Wo=P, Cout=Cc, kh=1, kw=1, stride=1,
               pad=0, transposed=False, prologue=3 if inverse else 2,
               epilogue=L.EPI_IGDN_BWD if inverse else L.EPI_GDN_BWD, aux=g, aux2=xh, aux3=norm, out2=t)
    else:
        L.check(lib.lic_gdn_dnorm(_ptr(g), _ptr(xh), _ptr(norm), _ptr(t), xh.numel(), int(inverse),
                                  _stream()), "lic_gdn_dnorm")
        if need_dx:
            dxh = torch.empty_like(xh)
            _igemm(t, gp, dxh, B=1, Hi=1, Wi=P, Cin=Cc, Ho=1, Wo=P, Cout=Cc, kh=1, kw=1,
                   stride=1, pad=0, transposed=False, epilogue=L.EPI_IGDN_BWD if inverse else L.EPI_GDN_BWD,
                   aux=g, aux2=xh, aux3=norm)
    dge = None
    if dfr:
        # pending reductions that end in the re-parametrisation's backward: no lic_gdn_reparam_bwd2 launch here
        dbeta, dgamma = dbeta_d, None
        if need_dbeta and dbeta is None:
            dbeta = _colsum(t, P, Cc, deferred=True, reparam=(beta_c, beta_bound)).view(beta_c.shape)
        if need_dgamma:
            dgamma = torch.empty_like(gamma_c)
            job = L.ReduceJob()
            _wgrad(t, xh, dgamma, B=1, Hs=1, Ws=P, Cp=Cc, Hl=1, Wl=P, Cg=Cc, kh=1, kw=1, stride=1, pad=0,
                   g_is_row=False, dst_sm=Cc, dst_sn=1, dst_stap=0, sq_g=1, job=job)
            job.epilogue, job.param, job.bound = L.REDUCE_EPI_REPARAM, gamma_c.data_ptr(), gamma_bound
            _PENDING_KEEP.append(gamma_c)
        if cs_dx is not None and dxh is not None:
            dxh._lic_colsum_partial = cs_dx
        return dxh, dbeta, dgamma
    if need_dbeta and dbe is None:
        dbe = _colsum(t, P, Cc)
    if need_dgamma:
        dge = torch.empty_like(gamma_c)
        _wgrad(t, xh, dge, B=1, Hs=1, Ws=P, Cp=Cc, Hl=1, Wl=P, Cg=Cc, kh=1, kw=1, stride=1, pad=0,
               g_is_row=False, dst_sm=Cc, dst_sn=1, dst_stap=0, sq_g=1)
    dbeta, dgamma = _reparam_bwd2(beta_c if need_dbeta else None, dbe, beta_bound,
                                  gamma_c if need_dgamma else None, dge, gamma_bound)
    if cs_dx is not None and dxh is not None:
        dxh._lic_colsum_partial = cs_dx  # [rows][C] partial column sums of dx: the conv in front needs only these
    return dxh, dbeta, dgamma


class _ConvGDNFn(torch.autograd.Function):
    """conv / transposed conv followed by GDN / IGDN as ONE kernel launch (the pairs at
    Components.py:10-15 and :39-44): the tile that produced all channels of a pixel also pools them.
    Forward values are bitwise those of conv2d -> gdn; backward is the two ops' backward in sequence."""

    @staticmethod
    def forward(ctx, x, weight, bias, beta, gamma, stride, pad, out_pad, transposed, inverse, beta_bound,
                gamma_bound, pedestal, keep=True):
        _require_cuda(x, weight, bias, beta, gamma)
        xh = _nhwc(x)
        B, Hi, Wi, Cin = xh.shape
        kh, kw = weight.shape[2], weight.shape[3]
        Cout = weight.shape[1] if transposed else weight.shape[0]
        Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, transposed, out_pad)
        wp = _pack_conv_weight(weight, transposed, for_dgrad=False)
        beta_e, gT = _gdn_operands(beta, gamma, beta_bound, gamma_bound, pedestal)
        y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        conv_out = torch.empty_like(y) if keep else None   # (only the backward pass reads these two)
        norm = torch.empty_like(y) if keep else None
        _igemm(xh, wp, y, B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, Cout=Cout, kh=kh, kw=kw, stride=stride,
               pad=pad, transposed=transposed, bias=bias,
               epilogue=L.EPI_CONV_IGDN if inverse else L.EPI_CONV_GDN, out2=norm, out3=conv_out, aux=gT,
               aux2=beta_e)
        ctx.save_for_backward(xh, weight, conv_out, norm, beta, gamma)
        ctx.cfg = (stride, pad, transposed, inverse, beta_bound, gamma_bound, bias is not None, pedestal)
        return _nchw_view(y)

    @staticmethod
    def backward(ctx, gy):
        xh, weight, conv_out, norm, beta, gamma = ctx.saved_tensors
        stride, pad, transposed, inverse, beta_bound, gamma_bound, has_bias, pedestal = ctx.cfg
        need = ctx.needs_input_grad
        g_conv, dbeta, dgamma = _gdn_backward(conv_out, norm, beta, gamma, _nhwc(gy), inverse,
                                              beta_bound, gamma_bound, pedestal, True, need[3], need[4])
        dx, dw, db = _conv_backward(xh, weight, g_conv, stride, pad, transposed, 0, need[0], need[1],
                                    has_bias and need[2])
        return dx, dw, db, dbeta, dgamma, None, None, None, None, None, None, None, None, None


def will_backprop(*ts) -> bool:
    """True when autograd will record the op being built.  Inside Function.forward grad mode is always off and
    ctx.needs_input_grad ignores torch.no_grad(), so the wrappers ask before .apply: tensors that only the backward
    pass reads (the GDN norm, the convolution output in front of a fused GDN) are written only then."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def fused_gdn_supported(cin: int, cout: int) -> bool:
    return bool(L.load().lic_igemm_fused_gdn_supported(int(cin), int(cout)))


def fused_gdn_preferred(x_shape, weight_shape, stride, padding, transposed, output_padding=0) -> bool:
    """True when the one-launch conv+GDN is also expected to be faster than two launches for this
    geometry (lic_igemm_fused_gdn_preferred: layers that run on 64-row tiles anyway)."""
    B, Cin, Hi, Wi = x_shape
    kh, kw = weight_shape[2], weight_shape[3]
    Cout = weight_shape[1] if transposed else weight_shape[0]
    Ho, Wo = conv_out_size(Hi, Wi, kh, stride, padding, transposed, output_padding)
    d = L.IgemmDesc()
    if not transposed and Cin < 4:  # RGB stem: columns + dense GEMM
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = 1, 1, B * Ho * Wo, _kpad(kh, kw, Cin), 1, B * Ho * Wo, Cout
        d.kh = d.kw = d.stride = 1
        d.pad = d.transposed = 0
    else:
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, Hi, Wi, Cin, Ho, Wo, Cout
        d.kh, d.kw, d.stride, d.pad, d.transposed = kh, kw, stride, padding, int(transposed)
    d.in_ld, d.out_ld, d.out2_ld, d.out3_ld = d.Cin, Cout, Cout, Cout
    return bool(L.load().lic_igemm_fused_gdn_preferred(C.byref(d)))


def conv_gdn(x, weight, bias, beta, gamma, stride, padding, inverse, beta_bound, gamma_bound, pedestal=PEDESTAL,
             transposed=False, output_padding=0):
    """`gdn(conv2d(x))` / `gdn(conv_transpose2d(x))` in one launch (see _ConvGDNFn)."""
    if not transposed and weight.shape[1] < 4:  # RGB stem: columns + dense GEMM
        return _ImageConvGDNFn.apply(x, weight, bias, beta, gamma, stride, padding, bool(inverse),
                                     float(beta_bound), float(gamma_bound), float(pedestal),
                                     will_backprop(x, weight, bias, beta, gamma))
    return _ConvGDNFn.apply(x, weight, bias, beta, gamma, stride, padding, output_padding, bool(transposed),
                            bool(inverse), float(beta_bound), float(gamma_bound), float(pedestal),
                            will_backprop(x, weight, bias, beta, gamma))


def gdn(x, beta, gamma, inverse, beta_bound, gamma_bound, pedestal=PEDESTAL, residual=None):
    return _GDNFn.apply(x, beta, gamma, bool(inverse), float(beta_bound), float(gamma_bound),
                        float(pedestal), residual, will_backprop(x, beta, gamma, residual))


# ------------------------------------------------------------------------------------------
# quantisation surrogate (Models.py:55-64)
# ------------------------------------------------------------------------------------------
_QUANT_CASTS = [None, None]   # (bf16 of the input, bf16 of the result) of the _QuantizeFn.forward that just ran


class _QuantizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, u, training, cast_in=False, cast_out=False):
        _require_cuda(v, u)
        vh = _nhwc(v)
        uh = None if u is None else _nhwc(u)
        out = torch.empty_like(vh)
        _QUANT_CASTS[0] = _QUANT_CASTS[1] = None
        if (cast_in or cast_out) and vh.numel() % 4 == 0:
            # the bf16 copies the bf16-storage consumers of v / of the result would each make with a cast launch
            v16 = torch.empty_like(vh, dtype=torch.bfloat16) if cast_in else None
            o16 = torch.empty_like(vh, dtype=torch.bfloat16) if cast_out else None
            L.check(L.load().lic_quantize_bf16(_ptr(vh), _ptr(uh), _ptr(out), _ptr(v16), _ptr(o16), vh.numel(),
                                               int(training), _stream()), "lic_quantize_bf16")
            _QUANT_CASTS[0], _QUANT_CASTS[1] = v16, o16
        else:
            L.check(L.load().lic_quantize(_ptr(vh), _ptr(uh), _ptr(out), vh.numel(), int(training), _stream()),
                    "lic_quantize")
        ctx.training = training
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, g):
        # additive noise: identity; round(): zero gradient (as torch.round)
        return (g if ctx.training else torch.zeros_like(g)), None, None, None, None


def attach_bf16(t: torch.Tensor, t16_nhwc: Optional[torch.Tensor]):
    """remember the bf16 NHWC copy of `t` that some launch already wrote, for functional_bf16's input casts (valid while
    `t` is not modified in place: the version counter is checked on use)"""
    if t16_nhwc is not None:
        t._lic_bf16 = (t._version, t16_nhwc)


def quantize(v, u=None, training=True, cast_in=False, cast_out=False):
    """`cast_in` / `cast_out`: the same launch also writes bf16(v) / bf16(result) and attaches them to `v` / the result
    (attach_bf16): the bf16-storage layers that consume them skip their cast launches"""
    if not (cast_in or cast_out):
        return _QuantizeFn.apply(v, u, training)
    res = _QuantizeFn.apply(v, u, training, cast_in, cast_out)
    attach_bf16(v, _QUANT_CASTS[0])
    attach_bf16(res, _QUANT_CASTS[1])
    _QUANT_CASTS[0] = _QUANT_CASTS[1] = None
    return res


# ------------------------------------------------------------------------------------------
# entropy-parameter activations + conditional likelihood
# ------------------------------------------------------------------------------------------
class _EntropyParamsActFn(torch.autograd.Function):
    """softplus(.)+1e-6 on scales, softmax over K on weights (ParametersModels.py:43-64)."""

    @staticmethod
    def forward(ctx, raw, M, K):
        _require_cuda(raw)
        rh = _nhwc(raw)
        B, H, W, CH = rh.shape
        out = torch.empty_like(rh)
        L.check(L.load().lic_entropy_params_fwd(_ptr(rh), _ptr(out), B * H * W, M, K, _stream()),
                "lic_entropy_params_fwd")
        ctx.save_for_backward(rh, out)
        ctx.cfg = (M, K)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, g):
        rh, out = ctx.saved_tensors
        M, K = ctx.cfg
        gh = _nhwc(g)
        B, H, W, CH = rh.shape
        d = torch.empty_like(rh)
        L.check(L.load().lic_entropy_params_bwd(_ptr(rh), _ptr(out), _ptr(gh), _ptr(d), B * H * W, M, K,
                                                _stream()), "lic_entropy_params_bwd")
        return _nchw_view(d), None, None


def entropy_params_activation(raw, M, K):
    return _EntropyParamsActFn.apply(raw, M, K)


class _GmmLikelihoodFn(torch.autograd.Function):
    """p = clamp_min(sum_k w_k [Phi((x+.5-mu_k)/s_k) - Phi((x-.5-mu_k)/s_k)], bound); logp = log p
    (EntropyModels.py:29-31,188-233; Models.py:86-87).  params: [B, G*K*M, h, w]."""

    @staticmethod
    def forward(ctx, x, params, K, bound):
        _require_cuda(x, params)
        ctx.set_materialize_grads(False)   # (p is not part of the loss: no zero-filled gradient for it)
        xh, ph = _nhwc(x), _nhwc(params)
        B, H, W, M = xh.shape
        p = torch.empty_like(xh)
        logp = torch.empty_like(xh)
        L.check(L.load().lic_gmm_likelihood_fwd(_ptr(xh), _ptr(ph), _ptr(p), _ptr(logp), B * H * W, M, K,
                                                bound, _stream()), "lic_gmm_likelihood_fwd")
        ctx.save_for_backward(xh, ph)
        ctx.cfg = (K, bound)
        return _nchw_view(p), _nchw_view(logp)

    @staticmethod
    def backward(ctx, gp, glogp):
        xh, ph = ctx.saved_tensors
        K, bound = ctx.cfg
        B, H, W, M = xh.shape
        gph = None if gp is None else _nhwc(gp)
        glh = None if glogp is None else _nhwc(glogp)
        dx = torch.empty_like(xh)
        dpar = torch.empty_like(ph)
        L.check(L.load().lic_gmm_likelihood_bwd(_ptr(xh), _ptr(ph), _ptr(gph), _ptr(glh), _ptr(dx),
                                                _ptr(dpar), B * H * W, M, K, bound, _stream()),
                "lic_gmm_likelihood_bwd")
        return _nchw_view(dx), _nchw_view(dpar), None, None


def gmm_likelihood(x, params, K, bound=LIKELIHOOD_BOUND):
    return _GmmLikelihoodFn.apply(x, params, K, bound)


# ------------------------------------------------------------------------------------------
# factorised bottleneck
# ------------------------------------------------------------------------------------------
_FE_SIZES = [3, 9, 9, 3, 3, 3, 3, 1, 3, 3, 3]


def _fe_pack(plist, Cc):
    """[C][43] operand of the factorised kernels from the 11 parameter tensors: one gather launch when they are the
    reference's contiguous fp32 tensors, torch.cat otherwise"""
    if len(plist) == 11 and all(q.is_contiguous() and q.dtype == torch.float32 and q.shape[0] == Cc for q in plist) and \
            [q.shape[1] * q.shape[2] for q in plist] == _FE_SIZES:
        packed = torch.empty((Cc, 43), device=plist[0].device, dtype=torch.float32)
        ptrs = (C.c_void_p * 11)(*[q.data_ptr() for q in plist])
        L.check(L.load().lic_fe_pack(ptrs, _ptr(packed), Cc, _stream()), "lic_fe_pack")
        return packed
    return torch.cat([q.reshape(Cc, -1) for q in plist], dim=1).contiguous()


class _FactorizedFn(torch.autograd.Function):
    """EntropyModels.py:49-151 (+ clamp :29-31, log Models.py:83-84)."""

    @staticmethod
    def forward(ctx, x, bound, *plist):
        _require_cuda(x, *plist)
        ctx.set_materialize_grads(False)   # (p is not part of the loss: no zero-filled gradient for it)
        Cc = plist[0].shape[0]
        packed = _fe_pack(plist, Cc)
        if x.dim() == 4:
            xh = _nhwc(x)
        else:  # (B, C) or (B, C, N): bring channels last
            xh = x.reshape(x.shape[0], Cc, -1, 1).permute(0, 2, 3, 1).contiguous()
        P = xh.numel() // Cc
        p = torch.empty_like(xh)
        logp = torch.empty_like(xh)
        L.check(L.load().lic_factorized_fwd(_ptr(xh), _ptr(packed), _ptr(p), _ptr(logp), P, Cc, bound,
                                            _stream()), "lic_factorized_fwd")
        ctx.save_for_backward(xh, packed)
        ctx.cfg = (bound, tuple(x.shape), [tuple(q.shape) for q in plist])
        if x.dim() == 4:
            return _nchw_view(p), _nchw_view(logp)
        back = lambda t: t.permute(0, 3, 1, 2).reshape(x.shape)
        return back(p), back(logp)

    @staticmethod
    def backward(ctx, gp, glogp):
        xh, packed = ctx.saved_tensors
        bound, xshape, pshapes = ctx.cfg
        Cc = packed.shape[0]
        P = xh.numel() // Cc

        def to_h(g):
            if g is None:
                return None
            if len(xshape) == 4:
                return _nhwc(g)
            return g.reshape(xshape[0], Cc, -1, 1).permute(0, 2, 3, 1).contiguous()
        gph, glh = to_h(gp), to_h(glogp)
        dx = torch.empty_like(xh)
        dpk = torch.empty_like(packed)
        L.check(L.load().lic_factorized_bwd(_ptr(xh), _ptr(packed), _ptr(gph), _ptr(glh), _ptr(dx), _ptr(dpk),
                                            P, Cc, bound, _stream()), "lic_factorized_bwd")
        dxo = _nchw_view(dx) if len(xshape) == 4 else dx.permute(0, 3, 1, 2).reshape(xshape)
        if len(pshapes) == 11 and [s_[1] * s_[2] for s_ in pshapes] == _FE_SIZES:
            # one launch: the [C][43] gradient scattered parameter-major; each block IS a parameter's gradient
            flat = torch.empty((Cc * 43,), device=dpk.device, dtype=torch.float32)
            L.check(L.load().lic_fe_unpack(_ptr(dpk), _ptr(flat), Cc, _stream()), "lic_fe_unpack")
            grads, off = [], 0
            for n_, s_ in zip(_FE_SIZES, pshapes):
                grads.append(flat[off:off + Cc * n_].view(s_))
                off += Cc * n_
        else:
            grads = [g.reshape(s) for g, s in zip(torch.split(dpk, _FE_SIZES, dim=1), pshapes)]
        return (dxo, None, *grads)


def factorized_likelihood(x, matrices, biases, factors, bound=LIKELIHOOD_BOUND):
    return _FactorizedFn.apply(x, bound, *matrices, *biases, *factors)


def factorized_channel_logits(matrices, biases, factors, ch: int, xs: torch.Tensor) -> torch.Tensor:
    _require_cuda(xs)
    Cc = matrices[0].shape[0]
    with torch.no_grad():
        packed = torch.cat([q.reshape(Cc, -1) for q in (*matrices, *biases, *factors)], dim=1).contiguous()
        xs_c = xs.contiguous()
        out = torch.empty_like(xs_c)
        L.check(L.load().lic_factorized_channel_logits(_ptr(packed), int(ch), _ptr(xs_c), _ptr(out),
                                                       xs_c.numel(), _stream()),
                "lic_factorized_channel_logits")
    return out


# ------------------------------------------------------------------------------------------
# rate-distortion loss
# ------------------------------------------------------------------------------------------
class _RdLossFn(torch.autograd.Function):
    """RateDistortionLoss.py:5-49 as one fused reduction; returns the small result buffer."""

    @staticmethod
    def forward(ctx, logp_y, logp_z, x_hat, x, lambda_rd):
        _require_cuda(logp_y, logp_z, x_hat, x)
        ly, lz = _nhwc(logp_y), _nhwc(logp_z)
        xh, xx = _nhwc(x_hat), _nhwc(x)
        B = xx.shape[0]
        lib = L.load()
        out = torch.empty((16 + 2 * B,), device=x.device, dtype=torch.float32)   # (lic_rd_loss_fwd writes every slot)
        nbytes = lib.lic_rd_loss_workspace_bytes(B)
        ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
        num_pixels = xx.shape[1] * xx.shape[2]
        L.check(lib.lic_rd_loss_fwd(_ptr(ly), ly.numel() // B, _ptr(lz), lz.numel() // B, _ptr(xh), _ptr(xx),
                                    xx.numel() // B, B, num_pixels, lambda_rd, _ptr(out), _ptr(ws), nbytes,
                                    _stream()), "lic_rd_loss_fwd")
        ctx.save_for_backward(xh, xx)
        ctx.cfg = (tuple(ly.shape), tuple(lz.shape), lambda_rd, num_pixels)
        # the loss as a tensor of its own on element 0 of the buffer: a `buf[0]` in the caller would make every
        # backward start with select_backward's zero-fill and a 4-byte device copy (a memcpy node, which a launch
        # plan -- plan.StepPlan -- cannot read back from the captured graph)
        loss = out.new_empty(()).set_(out.untyped_storage(), out.storage_offset(), ())
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)   # (no zero-fill launch for the buffer's never-used gradient)
        return out, loss

    @staticmethod
    def backward(ctx, _gbuf, gloss):
        xh, xx = ctx.saved_tensors
        shy, shz, lam, num_pixels = ctx.cfg
        B = xx.shape[0]
        if gloss is None:
            return None, None, None, None, None
        gl = gloss.reshape(1).contiguous()  # only `loss` carries gradient
        dly = torch.empty(shy, device=xx.device, dtype=torch.float32)
        dlz = torch.empty(shz, device=xx.device, dtype=torch.float32)
        dxh = torch.empty_like(xh)
        L.check(L.load().lic_rd_loss_bwd(_ptr(xh), _ptr(xx), dly.numel() // B, dlz.numel() // B,
                                         xx.numel() // B, B, num_pixels, lam, _ptr(gl), _ptr(dly), _ptr(dlz),
                                         _ptr(dxh), _stream()), "lic_rd_loss_bwd")
        return _nchw_view(dly), _nchw_view(dlz), _nchw_view(dxh), None, None


def rd_loss_buffer(logp_y, logp_z, x_hat, x, lambda_rd):
    """(result buffer [16 + 2B] -- no gradient --, loss: 0-d tensor on its element 0, differentiable)"""
    return _RdLossFn.apply(logp_y, logp_z, x_hat, x, float(lambda_rd))


def mask_weight_(weight: torch.Tensor, mask: torch.Tensor):
    """weight.data *= mask, in place (ContextModels.py:19)."""
    _require_cuda(weight, mask)
    L.check(L.load().lic_mul_inplace(_ptr(weight.data), _ptr(mask), weight.numel(), _stream()),
            "lic_mul_inplace")


# ------------------------------------------------------------------------------------------
# evaluation metric: MS-SSIM as Evaluator.py:38,45 calls pytorch-msssim 0.2.1
# ------------------------------------------------------------------------------------------
def ms_ssim(X: torch.Tensor, Y: torch.Tensor, data_range: float = 255.0, size_average: bool = True):
    """`pytorch_msssim.ms_ssim(X, Y, data_range, size_average)` (default window / weights / K) on the
    device.  X, Y: [B,C,H,W] fp32 CUDA tensors (any strides), min(H,W) > 160.  Returns the mean over
    batch and channels, or the per-image channel mean when `size_average=False`.  No gradient."""
    _require_cuda(X, Y)
    if X.shape != Y.shape or X.dim() != 4:
        raise ValueError(f"Input images should have the same 4-d shape, got {tuple(X.shape)} and {tuple(Y.shape)}")
    B, Cc, H, W = X.shape
    if min(H, W) <= 160:
        raise ValueError("Image size should be larger than 160 due to the 4 downsamplings in ms-ssim")
    Xd = X.detach().float()
    Yd = Y.detach().float()
    if Yd.stride() != Xd.stride():
        Yd = Yd.contiguous(memory_format=torch.channels_last if Xd.is_contiguous(memory_format=torch.channels_last)
                           and not Xd.is_contiguous() else torch.contiguous_format)
        if Yd.stride() != Xd.stride():
            Xd, Yd = Xd.contiguous(), Yd.contiguous()
    lib = L.load()
    nbytes = lib.lic_msssim_workspace_bytes(B, Cc, H, W)
    ws = torch.empty((nbytes + 7) // 8, device=X.device, dtype=torch.float64)
    out = torch.empty((B, Cc), device=X.device, dtype=torch.float32)
    levels = torch.empty((5, B * Cc, 2), device=X.device, dtype=torch.float32)
    sb, sc, sh, sw = Xd.stride()
    L.check(lib.lic_msssim(_ptr(Xd), _ptr(Yd), B, Cc, H, W, sb, sc, sh, sw, float(data_range), _ptr(out),
                           _ptr(levels), _ptr(ws), nbytes, _stream()), "lic_msssim")
    return out.mean() if size_average else out.mean(1)
