"""bf16-storage variants of the conv / GDN autograd Functions (BASELINE config 3).

Activations between the layers of the analysis / synthesis stacks are bf16 (NHWC), parameters stay
fp32 (the state dict is unchanged) and are packed to bf16 per call, every accumulation and every
parameter gradient is fp32.  The latent side (y, z, hyper path, entropy parameters, likelihoods,
rd_loss) stays fp32: the first layer of a stack casts its input, the last one writes fp32.
No CPU fallback, same rules as functional.py.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from .functional import (PEDESTAL, _colsum, _nchw_view, _nhwc, _permute3, _ptr, _reparam_bwd2, _stream, conv_out_size,
                         grad_like, prepared)

BF16 = torch.bfloat16
_NAMES = {}      # geometry -> kernel variant name (bench.py's event brackets)
_SPLIT_WS = {}   # geometry -> K-split workspace bytes of lic_igemm_bf16 (0 = no split)


def _check(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.LicError("neural_image_compression_amd runs on MI355X only: got a non-CUDA tensor "
                             "(there is no CPU fallback)")


def _as_bf16_nhwc(t: torch.Tensor) -> torch.Tensor:
    pre = getattr(t, "_lic_bf16", None)   # (functional.attach_bf16: a launch that already wrote this copy)
    if pre is not None and pre[0] == t._version and pre[1].shape == (t.shape[0], t.shape[2], t.shape[3], t.shape[1]):
        return pre[1]
    th = _nhwc(t)
    return th if th.dtype == BF16 else th.to(BF16)


def as_bf16(t: torch.Tensor) -> torch.Tensor:
    """the bf16 view of an activation the bf16 layers hand around (NHWC memory, NCHW-logical)"""
    return t if t.dtype == BF16 else t.to(BF16)


def _pack_bf16(src: torch.Tensor, taps, K, N, s_tap, s_k, s_n, kperm=False) -> torch.Tensor:
    lib = L.load()
    out = torch.empty((lib.lic_packed_weight_bf16_elems(taps, K, N),), device=src.device, dtype=BF16)
    fn = lib.lic_pack_weight_bf16_kperm if kperm else lib.lic_pack_weight_bf16
    L.check(fn(_ptr(src), _ptr(out), taps, K, N, s_tap, s_k, s_n, _stream()), "lic_pack_weight_bf16")
    return out


def _pack_conv_weight_bf16(w, transposed_weight, for_dgrad):
    hit = prepared(w, "bf16.dgrad" if for_dgrad else "bf16.fwd")
    if hit is not None:
        return hit
    w = w.contiguous()
    d0, d1, kh, kw = w.shape
    taps = kh * kw
    if transposed_weight:
        cin, cout, s_ci, s_co = d0, d1, d1 * taps, taps
    else:
        cout, cin, s_co, s_ci = d0, d1, d1 * taps, taps
    if for_dgrad:
        return _pack_bf16(w, taps, cout, cin, 1, s_co, s_ci)
    return _pack_bf16(w, taps, cin, cout, 1, s_ci, s_co)


def _igemm_bf16(inp, w_packed, out, *, B, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, transposed, bias=None,
                prologue=0, epilogue=L.EPI_NONE, out2=None, aux=None, aux2=None, aux3=None, slope=0.01, tap_mask=0,
                out3=None, out_ld=None):
    d = L.IgemmDesc()
    d.in_, d.w, d.bias, d.out, d.out2 = _ptr(inp), _ptr(w_packed), _ptr(bias), _ptr(out), _ptr(out2)
    d.aux, d.aux2, d.aux3, d.res, d.out3 = _ptr(aux), _ptr(aux2), _ptr(aux3), None, _ptr(out3)
    d.in_ld, d.out_ld = Cin, (Cout if out_ld is None else out_ld)
    d.out2_ld = d.aux_ld = d.aux2_ld = d.aux3_ld = d.res_ld = d.out3_ld = Cout
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, Hi, Wi, Cin, Ho, Wo, Cout
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.transposed, d.prologue, d.epilogue = int(transposed), prologue, epilogue
    d.tap_mask, d.slope = tap_mask, slope
    from . import functional as F_
    if F_.FORCE_IGEMM is not None:
        d.force_bm = F_.FORCE_IGEMM[0]  # the N tile follows from the channel count on this path
        d.force_split = F_.FORCE_IGEMM[2]
    if epilogue in (L.EPI_NONE, L.EPI_LEAKY) and prologue == 0 and out2 is None and \
            (Ho * Wo <= 192 or d.force_split > 1):
        # layers far too small to fill the chip split K across workgroups (fp32 partial tiles + a finishing launch);
        # the plan depends on the geometry only, so its workspace size is asked once per shape
        key = (B, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, bool(transposed), tap_mask, d.force_split, d.force_bm)
        nbytes = _SPLIT_WS.get(key)
        if nbytes is None:
            nbytes = _SPLIT_WS[key] = L.load().lic_igemm_bf16_workspace_bytes(C.byref(d))
        if nbytes:
            ws = torch.empty((nbytes // 4,), device=out.device, dtype=torch.float32)
            d.workspace, d.workspace_bytes = _ptr(ws), nbytes
    if F_.KERNEL_TRACE is not None:
        F_.KERNEL_TRACE.add(F_._kernel_name(L.load().lic_igemm_bf16_kernel_name, d))
    if F_.PROFILE is None or 2.0 * B * Ho * Wo * Cout * Cin * kh * kw < F_.PROFILE_MIN_FLOP:
        L.check(L.load().lic_igemm_bf16(C.byref(d), int(out.dtype == torch.float32), _stream()), "lic_igemm_bf16")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(L.load().lic_igemm_bf16(C.byref(d), int(out.dtype == torch.float32), _stream()), "lic_igemm_bf16")
    e1.record()
    if transposed and stride == 2:  # live taps per output pixel: kh*kw/4 on average
        macs = B * Ho * Wo * (kh * kw) * Cin * Cout // 4
    else:
        macs = B * Ho * Wo * kh * kw * Cin * Cout
    nkey = (B, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, bool(transposed), prologue, epilogue, tap_mask, d.force_bm)
    name = _NAMES.get(nkey)   # (planning the launch a second time for its name costs as much as the launch)
    if name is None:
        name = _NAMES[nkey] = F_._kernel_name(L.load().lic_igemm_bf16_kernel_name, d)
    F_.PROFILE.append((name, 2 * macs,
                       2 * B * Hi * Wi * Cin + out.element_size() * B * Ho * Wo * Cout, e0, e1))


def _wgrad_bf16(p, g, dst, *, B, Hs, Ws, Cp, Hl, Wl, Cg, kh, kw, stride, pad, g_is_row, dst_sm, dst_sn, dst_stap,
                sq_g=0, job=None):
    """`job` (an L.ReduceJob): launch the MFMA kernel only and fill `job` with the slab reduction, for
    functional.defer (the caller may still set its epilogue / index-map fields); None: reduce right away"""
    d = L.WgradDesc()
    d.p, d.g, d.dst = _ptr(p), _ptr(g), _ptr(dst)
    d.p_ld, d.g_ld = Cp, Cg
    d.dst_sm, d.dst_sn, d.dst_stap = dst_sm, dst_sn, dst_stap
    d.B, d.Hs, d.Ws, d.Cp, d.Hl, d.Wl, d.Cg = B, Hs, Ws, Cp, Hl, Wl, Cg
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.g_is_row, d.sq_p, d.sq_g, d.scale = int(g_is_row), 0, sq_g, 1.0
    lib = L.load()
    nbytes = lib.lic_wgrad_bf16_workspace_bytes(C.byref(d))
    ws = torch.empty((max(nbytes, 4) + 3) // 4, device=p.device, dtype=torch.float32)
    from . import functional as F_
    if F_.KERNEL_TRACE is not None:
        F_.KERNEL_TRACE.add(F_._kernel_name(lib.lic_wgrad_bf16_kernel_name, d))
    def launch():
        if job is None:
            L.check(lib.lic_wgrad_bf16(C.byref(d), _ptr(ws), nbytes, _stream()), "lic_wgrad_bf16")
        else:
            L.check(lib.lic_wgrad_bf16_partial(C.byref(d), _ptr(ws), nbytes, C.byref(job), _stream()),
                    "lic_wgrad_bf16_partial")
            F_.defer(job, ws, p, g)   # (not `dst`: autograd adopts a gradient tensor only if nobody else holds it)
    if F_.PROFILE is None or 2.0 * B * Hs * Ws * kh * kw * Cp * Cg < F_.PROFILE_MIN_FLOP:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    nkey = ("w", B, Hs, Ws, Cp, Hl, Wl, Cg, kh, kw, stride, pad, bool(g_is_row), sq_g)
    name = _NAMES.get(nkey)
    if name is None:
        name = _NAMES[nkey] = F_._kernel_name(lib.lic_wgrad_bf16_kernel_name, d) + "+reduce"
    if job is not None:
        name = name[:-len("+reduce")]
    F_.PROFILE.append((name, 2 * B * Hs * Ws * kh * kw * Cp * Cg,
                       2 * (B * Hs * Ws * Cp + B * Hl * Wl * Cg), e0, e1))


def _reparam_epilogue(job, param_c, bound):
    job.epilogue, job.param, job.bound = L.REDUCE_EPI_REPARAM, param_c.data_ptr(), bound


def _colsum_bf16(t2d, P, Cc, defer=False, reparam=None):
    """column sums; `defer`: stage 2 joins the backward pass's batched reduction (functional.defer); `reparam` =
    (parameter, bound): followed by the GDN re-parametrisation's backward (deferred mode only)"""
    lib = L.load()
    nbytes = lib.lic_colsum_bf16_workspace_bytes(P, Cc)
    ws = torch.empty((nbytes + 3) // 4, device=t2d.device, dtype=torch.float32)
    out = torch.empty((Cc,), device=t2d.device, dtype=torch.float32)
    if defer:
        from . import functional as F_
        job = L.ReduceJob()
        L.check(lib.lic_colsum_bf16_partial(_ptr(t2d), Cc, P, Cc, 1.0, _ptr(out), _ptr(ws), nbytes, C.byref(job), _stream()),
                "lic_colsum_bf16_partial")
        if reparam is not None:
            _reparam_epilogue(job, *reparam)
        F_.defer(job, ws, t2d, reparam[0] if reparam is not None else None)
        return out
    L.check(lib.lic_colsum_bf16(_ptr(t2d), Cc, P, Cc, 1.0, _ptr(out), _ptr(ws), nbytes, _stream()),
            "lic_colsum_bf16")
    return out


def _rows_sum(part, defer=False, reparam=None):
    """column sums of a small fp32 [rows][C] matrix of partial sums (the per-workgroup rows lic_gdn_bwd_bf16 leaves):
    pending (`defer`: one COLUMNS job of the pass's batched reduction, optionally with the re-parametrisation's
    backward) or right away (a one-job lic_reduce_batch)"""
    from . import functional as F_
    rows, Cc = part.shape
    out = torch.empty((Cc,), device=part.device, dtype=torch.float32)
    job = L.ReduceJob()
    job.src, job.dst, job.kind, job.splitk, job.Cn, job.scale = part.data_ptr(), out.data_ptr(), L.REDUCE_COLUMNS, rows, Cc, 1.0
    if not defer:   # the same kernel, now (so that deferring changes no bit)
        L.check(L.load().lic_reduce_batch(C.byref(job), 1, _stream()), "lic_reduce_batch")
        return out
    if reparam is not None:
        _reparam_epilogue(job, *reparam)
    F_.defer(job, part, reparam[0] if reparam is not None else None)
    return out


def _colsum2_bf16(a2d, b2d, P, Cc, defer=False, reparam_a=None):
    """column sums of two bf16 [P][Cc] matrices in one launch pair (`defer` / `reparam_a`: as _colsum_bf16, the
    re-parametrisation applies to the first matrix's sums)"""
    lib = L.load()
    nbytes = 2 * lib.lic_colsum_bf16_workspace_bytes(P, Cc)
    ws = torch.empty((nbytes + 3) // 4, device=a2d.device, dtype=torch.float32)
    out = torch.empty((2, Cc), device=a2d.device, dtype=torch.float32)
    if defer:
        from . import functional as F_
        jobs = (L.ReduceJob * 2)()
        L.check(lib.lic_colsum2_bf16_partial(_ptr(a2d), _ptr(b2d), Cc, P, Cc, 1.0, _ptr(out[0]), _ptr(out[1]), _ptr(ws),
                                             nbytes, jobs, _stream()), "lic_colsum2_bf16_partial")
        ja, jb = L.ReduceJob.from_buffer_copy(jobs[0]), L.ReduceJob.from_buffer_copy(jobs[1])
        if reparam_a is not None:
            _reparam_epilogue(ja, *reparam_a)
        F_.defer(ja, ws, a2d, b2d, reparam_a[0] if reparam_a is not None else None)
        F_.defer(jb)
        return out[0], out[1]
    L.check(lib.lic_colsum2_bf16(_ptr(a2d), _ptr(b2d), Cc, P, Cc, 1.0, _ptr(out[0]), _ptr(out[1]), _ptr(ws), nbytes,
                                 _stream()), "lic_colsum2_bf16")
    return out[0], out[1]


def _leaky_bwd_bf16(y, dy, slope):
    dx = torch.empty_like(y)
    L.check(L.load().lic_leaky_bwd_bf16(_ptr(y), _ptr(dy), _ptr(dx), y.numel(), slope, _stream()), "lic_leaky_bwd_bf16")
    return dx


# the stem's column matrix for its weight gradient built beside the forward pass (functional.AUX_STREAM; 0: in the backward pass)
EARLY_STEM_COLUMNS = os.environ.get("LIC_EARLY_STEM_COLUMNS", "1") != "0"
# the bias gradient of a conv -> LeakyReLU layer out of the pass that masks its gradient (LIC_BF16_LEAKY_COLSUM=0: two passes)
LEAKY_COLSUM = os.environ.get("LIC_BF16_LEAKY_COLSUM", "1") != "0"


def _leaky_bwd_colsum_bf16(y, dy, slope, P, Cc, defer=False):
    """(dy through the LeakyReLU's backward, its column sums): _leaky_bwd_bf16 + _colsum_bf16 in one pass, the same bits"""
    lib = L.load()
    dx = torch.empty_like(y)
    nbytes = lib.lic_colsum_bf16_workspace_bytes(P, Cc)
    ws = torch.empty((nbytes + 3) // 4, device=y.device, dtype=torch.float32)
    out = torch.empty((Cc,), device=y.device, dtype=torch.float32)
    job = L.ReduceJob() if defer else None
    L.check(lib.lic_leaky_bwd_colsum_bf16(_ptr(y), _ptr(dy), _ptr(dx), P, Cc, slope, _ptr(out), _ptr(ws), nbytes,
                                          C.byref(job) if defer else None, _stream()), "lic_leaky_bwd_colsum_bf16")
    if defer:
        from . import functional as F_
        F_.defer(job, ws)
    return dx, out


class _ConvBF16Fn(torch.autograd.Function):
    """nn.Conv2d / nn.ConvTranspose2d with bf16 activations (Components.py:12-16,39-43), optionally with the
    LeakyReLU behind it fused (hyper / entropy-parameter layers: Components.py:69-73,99-103;
    ParametersModels.py:22-34) and a tap mask (ContextModels.py:19-20)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad, transposed, out_f32, leaky=False, slope=0.01, tap_mask=0,
                out_view=None):
        _check(x, weight, bias)
        xh = _as_bf16_nhwc(x)
        B, Hi, Wi, Cin = xh.shape
        kh, kw = weight.shape[2], weight.shape[3]
        Cout = weight.shape[1] if transposed else weight.shape[0]
        Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, transposed, out_pad)
        wp = _pack_conv_weight_bf16(weight, transposed, False)
        out_ld = None
        if out_view is not None:
            # the caller's channel range of a wider bf16 NHWC buffer (two producers fill one tensor: no torch.cat)
            if leaky or out_f32 or out_view.dtype != BF16 or tuple(out_view.shape) != (B, Ho, Wo, Cout) or \
                    out_view.stride(3) != 1 or out_view.stride(2) % 8 or out_view.data_ptr() % 16 or \
                    out_view.stride(1) != Wo * out_view.stride(2) or out_view.stride(0) != Ho * out_view.stride(1):
                raise ValueError("out_view must be a 16-byte aligned [B,Ho,Wo,Cout] channel slice of a contiguous bf16 "
                                 "NHWC buffer")
            out, out_ld = out_view, out_view.stride(2)
        else:
            out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32 if out_f32 else BF16)
        if leaky and out_f32:
            raise NotImplementedError("the fused LeakyReLU keeps its mask in the bf16 output")
        _igemm_bf16(xh, wp, out, B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, Cout=Cout, kh=kh, kw=kw, stride=stride,
                    pad=pad, transposed=transposed, bias=bias, epilogue=L.EPI_LEAKY if leaky else L.EPI_NONE,
                    slope=slope, tap_mask=tap_mask, out_ld=out_ld)
        ctx.save_for_backward(xh, weight, out if leaky else None)
        ctx.cfg = (stride, pad, transposed, bias is not None, x.dtype, leaky, slope, tap_mask)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, gy):
        xh, weight, yh = ctx.saved_tensors
        stride, pad, transposed, has_bias, in_dtype, leaky, slope, tap_mask = ctx.cfg
        g = _as_bf16_nhwc(gy)
        need = ctx.needs_input_grad
        dx, dw, db = _conv_backward_bf16(xh, weight, g, stride, pad, transposed, in_dtype, tap_mask, need[0], need[1],
                                         has_bias and need[2], leaky_y=yh if leaky else None, slope=slope)
        return dx, dw, db, None, None, None, None, None, None, None, None, None


def _conv_backward_bf16(xh, weight, g, stride, pad, transposed, in_dtype, tap_mask, need_dx, need_dw, need_db,
                        leaky_y=None, slope=0.01):
    """input / weight / bias gradients of a bf16-storage convolution from g = dL/d(conv output), bf16 NHWC.
    `leaky_y`: the layer's output went through a fused LeakyReLU (that output): g is the gradient behind it and passes
    the LeakyReLU's backward first -- in the same pass as the bias sum when there is one"""
    B, Hi, Wi, Cin = xh.shape
    _, Ho, Wo, Cout = g.shape
    kh, kw = weight.shape[2], weight.shape[3]
    taps = kh * kw
    dx = dw = db = None
    from . import functional as F_
    # the slab reduction of the weight gradient and the second stage of the bias sum wait for the end of the backward
    # pass when nothing can read these gradients earlier (functional.can_defer): one batched launch instead of ~40
    dfr = (need_dw or need_db) and F_.can_defer(weight)
    if leaky_y is not None:
        if need_db and LEAKY_COLSUM and Cout % 8 == 0 and g.is_contiguous() and leaky_y.is_contiguous():
            g, db = _leaky_bwd_colsum_bf16(leaky_y, g, slope, B * Ho * Wo, Cout, defer=dfr)
            need_db = False
        else:
            g = _leaky_bwd_bf16(leaky_y, g, slope)
    if need_dx:
        wp = _pack_conv_weight_bf16(weight, transposed, True)
        dxh = torch.empty((B, Hi, Wi, Cin), device=g.device, dtype=in_dtype)
        _igemm_bf16(g, wp, dxh, B=B, Hi=Ho, Wi=Wo, Cin=Cout, Ho=Hi, Wo=Wi, Cout=Cin, kh=kh, kw=kw, stride=stride,
                    pad=pad, transposed=not transposed, tap_mask=tap_mask)
        dx = _nchw_view(dxh)
    if need_dw:
        dw = grad_like(weight)
        job = L.ReduceJob() if dfr else None
        if transposed:
            _wgrad_bf16(xh, g, dw, B=B, Hs=Hi, Ws=Wi, Cp=Cin, Hl=Ho, Wl=Wo, Cg=Cout, kh=kh, kw=kw, stride=stride,
                        pad=pad, g_is_row=False, dst_sm=Cout * taps, dst_sn=taps, dst_stap=1, job=job)
        else:
            _wgrad_bf16(g, xh, dw, B=B, Hs=Ho, Ws=Wo, Cp=Cout, Hl=Hi, Wl=Wi, Cg=Cin, kh=kh, kw=kw, stride=stride,
                        pad=pad, g_is_row=True, dst_sm=taps, dst_sn=Cin * taps, dst_stap=1, job=job)
    if need_db:
        db = _colsum_bf16(g, B * Ho * Wo, Cout, defer=dfr)
    return dx, dw, db


def _kpad8(kh, kw, c):
    return (kh * kw * c + 7) // 8 * 8


class _ImageConvBF16Fn(torch.autograd.Function):
    """RGB stem (Components.py:10): fp32 image -> bf16 columns -> bf16 MFMA GEMM -> bf16 activations."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        _check(x, weight, bias)
        if x.requires_grad:
            raise NotImplementedError("the bf16 stem does not produce a gradient for the image")
        col, wpk, (B, Ho, Wo, Cout, Cin, Kp, P) = _stem_columns_bf16(x, weight, stride, pad)
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=BF16)
        _igemm_bf16(col, wpk, out, B=1, Hi=1, Wi=P, Cin=Kp, Ho=1, Wo=P, Cout=Cout,
                    kh=1, kw=1, stride=1, pad=0, transposed=False, bias=bias)
        ctx.save_for_backward(col, weight)
        ctx.cfg = (Cin, bias is not None)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, gy):
        col, weight = ctx.saved_tensors
        Cin, has_bias = ctx.cfg
        dw, db = _stem_backward_bf16(col, weight, _as_bf16_nhwc(gy), Cin, ctx.needs_input_grad[1],
                                     has_bias and ctx.needs_input_grad[2])
        return None, dw, db, None, None


def _stem_columns_bf16(x, weight, stride, pad):
    """fp32 image -> bf16 columns [P][Kp] plus the packed [Kp][Cout] weight of the RGB stem"""
    xh = _nhwc(x).float()
    B, Hi, Wi, Cin = xh.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, False)
    Kp, P, taps = _kpad8(kh, kw, Cin), B * Ho * Wo, kh * kw
    col = torch.empty((P, Kp), device=x.device, dtype=BF16)
    L.check(L.load().lic_im2col_bf16(_ptr(xh), _ptr(col), B, Hi, Wi, Cin, Ho, Wo, kh, kw, stride, pad, Kp, _stream()),
            "lic_im2col_bf16")
    wpk = prepared(weight, "bf16.stem")
    if wpk is None:
        wd = torch.zeros((Kp, Cout), device=x.device, dtype=torch.float32)
        _permute3(weight.contiguous(), wd, (taps, Cin, Cout), (1, taps, Cin * taps), (Cin * Cout, Cout, 1))
        wpk = _pack_bf16(wd, 1, Kp, Cout, 0, Cout, 1)
    return col, wpk, (B, Ho, Wo, Cout, Cin, Kp, P)


def _stem_backward_bf16(col, weight, g, Cin, need_dw, need_db):
    B, Ho, Wo, Cout = g.shape
    _, _, kh, kw = weight.shape
    taps, Kp, P = kh * kw, col.shape[1], B * Ho * Wo
    dw = db = None
    from . import functional as F_
    dfr = (need_dw or need_db) and F_.can_defer(weight)
    if need_dw:
        dw = grad_like(weight)
        if dfr:
            # the pending reduction writes dw[co][c][tap] itself: row m = tap * Cin + c of the [Kp][Cout] product goes to
            # tap + c * taps, column co to co * Cin * taps (what the permute launch below does otherwise)
            job = L.ReduceJob()
            _wgrad_bf16(col, g, dw, B=1, Hs=1, Ws=P, Cp=Kp, Hl=1, Wl=P, Cg=Cout, kh=1, kw=1, stride=1, pad=0,
                        g_is_row=False, dst_sm=Cout, dst_sn=1, dst_stap=0, job=job)
            job.mdiv, job.sm, job.smr, job.sn, job.Mvalid = Cin, 1, taps, Cin * taps, taps * Cin
        else:
            tmp = torch.empty((Kp, Cout), device=g.device, dtype=torch.float32)
            _wgrad_bf16(col, g, tmp, B=1, Hs=1, Ws=P, Cp=Kp, Hl=1, Wl=P, Cg=Cout, kh=1, kw=1, stride=1, pad=0,
                        g_is_row=False, dst_sm=Cout, dst_sn=1, dst_stap=0)
            _permute3(tmp, dw, (taps, Cin, Cout), (Cin * Cout, Cout, 1), (1, taps, Cin * taps))
    if need_db:
        db = _colsum_bf16(g, P, Cout, defer=dfr)
    return dw, db


def _head_direct(Cin, Cout, kh, kw, stride, pad, out_pad) -> bool:
    if os.environ.get("LIC_BF16_HEAD_DIRECT", "1") == "0":
        return False
    return bool(L.load().lic_head_convt_bf16_supported(Cin, Cout, kh, kw, stride, pad, out_pad))


class _ImageConvTBF16Fn(torch.autograd.Function):
    """RGB head (Components.py:45): bf16 activations -> bf16 per-tap columns -> fp32 image."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad):
        _check(x, weight, bias)
        xh = _as_bf16_nhwc(x)
        B, Hi, Wi, Cin = xh.shape
        _, Cout, kh, kw = weight.shape
        Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, True, out_pad)
        taps, Kp, P = kh * kw, _kpad8(kh, kw, Cout), B * Hi * Wi
        lib = L.load()
        wpk = prepared(weight, "bf16.head")
        if wpk is None:
            wd = torch.zeros((Cin, Kp), device=x.device, dtype=torch.float32)
            _permute3(weight.contiguous(), wd, (Cin, Cout, taps), (Cout * taps, taps, 1), (Kp, 1, Cout))
            wpk = _pack_bf16(wd, 1, Cin, Kp, 0, Kp, 1)
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        direct = _head_direct(Cin, Cout, kh, kw, stride, pad, out_pad)
        if direct:
            # features -> image in one launch, no column matrix (lic_head_bf16.hip)
            from . import functional as F_
            if F_.KERNEL_TRACE is not None:
                F_.KERNEL_TRACE.add(f"head_convt_bf16_kernel<{Cin // 16}>")
            F_._timed(f"head_convt_bf16_kernel<{Cin // 16}>", 2 * P * Cin * taps * Cout, 2 * P * Cin + 4 * B * Ho * Wo * Cout,
                      lambda: L.check(lib.lic_head_convt_bf16(_ptr(xh), _ptr(wpk), _ptr(bias), _ptr(out), B, Hi, Wi, Cin,
                                                              _stream()), "lic_head_convt_bf16"))
        else:
            col = torch.empty((P, Kp), device=x.device, dtype=BF16)
            _igemm_bf16(xh, wpk, col, B=1, Hi=1, Wi=P, Cin=Cin, Ho=1, Wo=P, Cout=Kp,
                        kh=1, kw=1, stride=1, pad=0, transposed=False)
            L.check(lib.lic_col2im_bf16(_ptr(col), _ptr(bias), _ptr(out), B, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad,
                                        Kp, _stream()), "lic_col2im_bf16")
        ctx.save_for_backward(xh, weight)
        ctx.cfg = (stride, pad, (Ho, Wo), bias is not None, x.dtype, direct)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, gy):
        xh, weight = ctx.saved_tensors
        stride, pad, (Ho, Wo), has_bias, in_dtype, direct = ctx.cfg
        g = _nhwc(gy).float()
        B, Hi, Wi, Cin = xh.shape
        _, Cout, kh, kw = weight.shape
        taps, Kp, P = kh * kw, _kpad8(kh, kw, Cout), B * Hi * Wi
        lib = L.load()
        dx = dw = db = None
        if ctx.needs_input_grad[0] and direct and in_dtype == BF16:
            # the data gradient = conv2d(dL/d image, the same [Cin][3][5][5] weight, stride 2, padding 2): one launch
            # from the image gradient, no column matrix (lic_stem_conv_bf16)
            wp16 = prepared(weight, "bf16.head_dx16")
            if wp16 is None:
                wp16 = torch.empty((lib.lic_stem_weight_bf16_elems(Cin),), device=g.device, dtype=BF16)
                L.check(lib.lic_pack_stem_weight_bf16(_ptr(weight.contiguous()), _ptr(wp16), Cin, _stream()),
                        "lic_pack_stem_weight_bf16")
            dxh = torch.empty((B, Hi, Wi, Cin), device=g.device, dtype=BF16)
            from . import functional as F_
            if F_.KERNEL_TRACE is not None:
                F_.KERNEL_TRACE.add(f"stem_gdn_bf16_kernel<{Cin // 32}, plain>")
            L.check(lib.lic_stem_conv_bf16(_ptr(g), _ptr(wp16), None, _ptr(dxh), B, Ho, Wo, Cin, _stream()),
                    "lic_stem_conv_bf16")
            dx = _nchw_view(dxh)
        need_dx_cols = ctx.needs_input_grad[0] and dx is None
        dcol = None
        if need_dx_cols or ctx.needs_input_grad[1]:
            dcol = torch.empty((P, Kp), device=g.device, dtype=BF16)
            L.check(lib.lic_im2col_bf16(_ptr(g), _ptr(dcol), B, Ho, Wo, Cout, Hi, Wi, kh, kw, stride, pad, Kp, _stream()),
                    "lic_im2col_bf16")
        if need_dx_cols:
            wpk = prepared(weight, "bf16.head_dx")
            if wpk is None:
                wdT = torch.zeros((Kp, Cin), device=g.device, dtype=torch.float32)
                _permute3(weight.contiguous(), wdT, (taps, Cout, Cin), (1, taps, Cout * taps), (Cout * Cin, Cin, 1))
                wpk = _pack_bf16(wdT, 1, Kp, Cin, 0, Cin, 1)
            dxh = torch.empty((B, Hi, Wi, Cin), device=g.device, dtype=in_dtype)
            _igemm_bf16(dcol, wpk, dxh, B=1, Hi=1, Wi=P, Cin=Kp, Ho=1, Wo=P,
                        Cout=Cin, kh=1, kw=1, stride=1, pad=0, transposed=False)
            dx = _nchw_view(dxh)
        if ctx.needs_input_grad[1]:
            from . import functional as F_
            dw = grad_like(weight)
            if F_.can_defer(weight):
                # pending reduction with the column -> (tap, colour) map: column n = tap * Cout + c of the [Cin][Kp] product
                # goes to tap + c * taps of dw[ci][c][tap]
                job = L.ReduceJob()
                _wgrad_bf16(xh, dcol, dw, B=1, Hs=1, Ws=P, Cp=Cin, Hl=1, Wl=P, Cg=Kp, kh=1, kw=1, stride=1, pad=0,
                            g_is_row=False, dst_sm=Kp, dst_sn=1, dst_stap=0, job=job)
                job.sm, job.ndiv, job.sn, job.snr, job.Nvalid = Cout * taps, Cout, 1, taps, taps * Cout
            else:
                tmp = torch.empty((Cin, Kp), device=g.device, dtype=torch.float32)
                _wgrad_bf16(xh, dcol, tmp, B=1, Hs=1, Ws=P, Cp=Cin, Hl=1, Wl=P, Cg=Kp, kh=1, kw=1, stride=1, pad=0,
                            g_is_row=False, dst_sm=Kp, dst_sn=1, dst_stap=0)
                _permute3(tmp, dw, (Cin, taps, Cout), (Kp, Cout, 1), (Cout * taps, 1, taps))
        if has_bias and ctx.needs_input_grad[2]:
            db = _colsum(g, B * Ho * Wo, Cout)
        return dx, dw, db, None, None, None


def _gamma_eff(gamma, gamma_bound, pedestal):
    gamma_c = gamma.contiguous()
    gamma_e = torch.empty_like(gamma_c)
    L.check(L.load().lic_gdn_reparam(_ptr(gamma_c), _ptr(gamma_e), gamma_c.numel(), gamma_bound, pedestal, _stream()),
            "lic_gdn_reparam")
    return gamma_e


def norm_recomputed_bf16(Cc: int) -> bool:
    """True when the GDN backward of this width recomputes the pool norm instead of reading it (lic_gdn_bwd_bf16_recompute):
    the forward pass then does not store it"""
    return os.environ.get("LIC_BF16_GDN_RECOMPUTE", "1") != "0" and os.environ.get("LIC_BF16_GDN_BWD", "1") != "0" and \
        bool(L.load().lic_gdn_bwd_bf16_supported(int(Cc)))


class _GDNBF16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, beta, gamma, inverse, beta_bound, gamma_bound, pedestal, keep=True):
        _check(x, beta, gamma)
        xh = _as_bf16_nhwc(x)
        B, H, W, Cc = xh.shape
        beta_e, gT = _gdn_operands_bf16(beta, gamma, beta_bound, gamma_bound, pedestal)
        out = torch.empty_like(xh)
        # (only the backward pass reads it -- and not even that where it recomputes the pool)
        norm = torch.empty_like(xh) if keep and not norm_recomputed_bf16(Cc) else None
        P = B * H * W
        _igemm_bf16(xh, gT, out, B=1, Hi=1, Wi=P, Cin=Cc, Ho=1, Wo=P, Cout=Cc, kh=1, kw=1, stride=1, pad=0,
                    transposed=False, bias=beta_e, prologue=1, epilogue=L.EPI_IGDN if inverse else L.EPI_GDN,
                    out2=norm, aux=xh)
        ctx.save_for_backward(xh, norm, beta, gamma)
        ctx.cfg = (inverse, beta_bound, gamma_bound, pedestal)
        return _nchw_view(out)

    @staticmethod
    def backward(ctx, gy):
        xh, norm, beta, gamma = ctx.saved_tensors
        need = ctx.needs_input_grad
        dxh, dbeta, dgamma = _gdn_backward_bf16(xh, norm, beta, gamma, _as_bf16_nhwc(gy), *ctx.cfg, need[0], need[1],
                                                need[2])
        return (None if dxh is None else _nchw_view(dxh)), dbeta, dgamma, None, None, None, None, None


def _gdn_operands_bf16(beta, gamma, beta_bound, gamma_bound, pedestal, kperm=False):
    """beta_eff (fp32) and gamma_eff^T packed as the bf16 operand of the pooling contraction (`kperm`: in the K
    order of the fused conv+GDN kernel, lic_pack_weight_bf16_kperm)"""
    beta_e, gT = prepared(beta, "f32.beta_e"), prepared(gamma, "bf16.gdn_gTp" if kperm else "bf16.gdn_gT")
    if beta_e is None or gT is None:
        Cc = beta.numel()
        beta_c = beta.contiguous()
        beta_e = torch.empty_like(beta_c)
        L.check(L.load().lic_gdn_reparam(_ptr(beta_c), _ptr(beta_e), Cc, beta_bound, pedestal, _stream()),
                "lic_gdn_reparam")
        gT = _pack_bf16(_gamma_eff(gamma, gamma_bound, pedestal), 1, Cc, Cc, 0, 1, Cc, kperm=kperm)
    return beta_e, gT


def _gdn_backward_bf16(xh, norm, beta, gamma, g, inverse, beta_bound, gamma_bound, pedestal, need_dx, need_dbeta,
                       need_dgamma, bias_from_dx=False):
    """gradients of y = x * norm^-1/2 (or ^1/2), norm = beta_eff + x^2 . gamma_eff^T, from g = dL/dy (bf16 NHWC);
    dx comes back as a bf16 NHWC tensor.  `bias_from_dx`: also return the column sums of dx -- the bias gradient of
    the convolution in front -- from the same launch pair as d-beta's (a 4th result)"""
    beta_c, gamma_c = beta.contiguous(), gamma.contiguous()
    lib = L.load()
    B, H, W, Cc = xh.shape
    P = B * H * W
    t = torch.empty_like(xh)
    dxh = dbeta = dgamma = None
    part_t = part_dx = None
    if norm is None and not (lib.lic_gdn_bwd_bf16_supported(Cc) and os.environ.get("LIC_BF16_GDN_BWD", "1") != "0"):
        raise L.LicError("GDN backward without a stored norm needs the recomputing kernel (C in {64, 128}); the forward "
                         "pass and the backward pass disagree about LIC_BF16_GDN_RECOMPUTE / LIC_BF16_GDN_BWD")
    if (need_dx or norm is None) and lib.lic_gdn_bwd_bf16_supported(Cc) and os.environ.get("LIC_BF16_GDN_BWD", "1") != "0":
        # one sweep: g, x, norm read once, t and dx written (lic_gdn_bf16.hip); the two-launch route below moves
        # 1.8x the bytes
        dxh = torch.empty_like(xh)
        gp = prepared(gamma, "bf16.gdn_gp")
        if gp is None:
            gp = _pack_bf16(_gamma_eff(gamma, gamma_bound, pedestal), 1, Cc, Cc, 0, Cc, 1, kperm=True)
        from . import functional as F_
        if F_.KERNEL_TRACE is not None:
            F_.KERNEL_TRACE.add(f"gdn_bwd_bf16_kernel<{Cc // 32}>")
        if need_dbeta and os.environ.get("LIC_BF16_GDN_BWD_CS", "1") != "0":
            # the kernel also leaves per-workgroup column sums of t and dx: d beta (and the convolution's d bias) need
            # one small reduction over those rows instead of a pass over the two activations
            rows = lib.lic_gdn_bwd_bf16_partial_rows(P)
            part_t = torch.empty((rows, Cc), device=xh.device, dtype=torch.float32)
            part_dx = torch.empty((rows, Cc), device=xh.device, dtype=torch.float32)
        if norm is None:
            # the forward pass did not store the pool: it is recomputed in the same sweep (two tensors read instead of three)
            beta_e, gTp = _gdn_operands_bf16(beta, gamma, beta_bound, gamma_bound, pedestal, kperm=True)
            F_._timed(f"gdn_bwd_bf16_kernel<{Cc // 32}>", 4 * P * Cc * Cc, 8 * P * Cc,
                      lambda: L.check(lib.lic_gdn_bwd_bf16_recompute(_ptr(g), _ptr(xh), _ptr(gp), _ptr(gTp), _ptr(beta_e),
                                                                     _ptr(dxh), _ptr(t), _ptr(part_t), _ptr(part_dx), P, Cc,
                                                                     int(inverse), _stream()), "lic_gdn_bwd_bf16_recompute"))
        else:
            F_._timed(f"gdn_bwd_bf16_kernel<{Cc // 32}>", 2 * P * Cc * Cc, 10 * P * Cc,
                      lambda: L.check(lib.lic_gdn_bwd_bf16(_ptr(g), _ptr(xh), _ptr(norm), _ptr(gp), _ptr(dxh), _ptr(t),
                                                           _ptr(part_t), _ptr(part_dx), P, Cc, int(inverse), _stream()),
                                      "lic_gdn_bwd_bf16"))
    else:
        L.check(lib.lic_gdn_dnorm_bf16(_ptr(g), _ptr(xh), _ptr(norm), _ptr(t), xh.numel(), int(inverse), _stream()),
                "lic_gdn_dnorm_bf16")
        if need_dx:
            dxh = torch.empty_like(xh)
            gp = prepared(gamma, "bf16.gdn_g")
            if gp is None:
                gp = _pack_bf16(_gamma_eff(gamma, gamma_bound, pedestal), 1, Cc, Cc, 0, Cc, 1)
            _igemm_bf16(t, gp, dxh, B=1, Hi=1, Wi=P, Cin=Cc, Ho=1, Wo=P, Cout=Cc,
                        kh=1, kw=1, stride=1, pad=0, transposed=False,
                        epilogue=L.EPI_IGDN_BWD if inverse else L.EPI_GDN_BWD, aux=g, aux2=xh, aux3=norm)
    dbe = dge = db_conv = None
    from . import functional as F_
    if (need_dbeta or need_dgamma) and F_.can_defer(beta if need_dbeta else None, gamma if need_dgamma else None):
        # pending reductions that end in the re-parametrisation's backward: d-beta, d-gamma (and the convolution's bias
        # gradient) come out of the backward pass's one batched launch, no lic_gdn_reparam_bwd2 launch here
        dbeta = dgamma = None
        if need_dbeta and part_t is not None:
            dbeta = _rows_sum(part_t, defer=True, reparam=(beta_c, beta_bound))
            if bias_from_dx:
                db_conv = _rows_sum(part_dx, defer=True)
        elif bias_from_dx and need_dbeta and dxh is not None:
            dbeta, db_conv = _colsum2_bf16(t, dxh, P, Cc, defer=True, reparam_a=(beta_c, beta_bound))
        elif need_dbeta:
            dbeta = _colsum_bf16(t, P, Cc, defer=True, reparam=(beta_c, beta_bound))
        if dbeta is not None:
            dbeta = dbeta.view(beta_c.shape)
        if need_dgamma:
            dgamma = torch.empty_like(gamma_c)
            job = L.ReduceJob()
            _wgrad_bf16(t, xh, dgamma, B=1, Hs=1, Ws=P, Cp=Cc, Hl=1, Wl=P, Cg=Cc, kh=1, kw=1, stride=1, pad=0,
                        g_is_row=False, dst_sm=Cc, dst_sn=1, dst_stap=0, sq_g=1, job=job)
            _reparam_epilogue(job, gamma_c, gamma_bound)
            F_._PENDING_KEEP.append(gamma_c)
        if bias_from_dx:
            return dxh, dbeta, dgamma, db_conv
        return dxh, dbeta, dgamma
    if need_dbeta and part_t is not None:
        dbe = _rows_sum(part_t)
        if bias_from_dx:
            db_conv = _rows_sum(part_dx)
    elif bias_from_dx and need_dbeta and dxh is not None:
        dbe, db_conv = _colsum2_bf16(t, dxh, P, Cc)
    elif need_dbeta:
        dbe = _colsum_bf16(t, P, Cc)
    if need_dgamma:
        dge = torch.empty_like(gamma_c)
        _wgrad_bf16(t, xh, dge, B=1, Hs=1, Ws=P, Cp=Cc, Hl=1, Wl=P, Cg=Cc, kh=1, kw=1, stride=1, pad=0,
                    g_is_row=False, dst_sm=Cc, dst_sn=1, dst_stap=0, sq_g=1)
    dbeta, dgamma = _reparam_bwd2(beta_c if need_dbeta else None, dbe, beta_bound,
                                  gamma_c if need_dgamma else None, dge, gamma_bound)
    if bias_from_dx:
        return dxh, dbeta, dgamma, db_conv
    return dxh, dbeta, dgamma


class _ConvGDNBF16Fn(torch.autograd.Function):
    """conv / transposed conv -> GDN / IGDN (Components.py:10-15, 39-44) as ONE launch in bf16 storage
    (LIC_EPI_CONV_GDN of lic_igemm_bf16).  Same rounding points as conv2d_bf16 -> gdn_bf16 (the conv output is
    bitwise the same; the pool sums every 16 channels in another order, so norm and y agree to fp32 / one-bf16-ulp
    rounding); the convolution output and the norm are written only when a backward pass will need them."""

    @staticmethod
    def forward(ctx, x, weight, bias, beta, gamma, stride, pad, out_pad, transposed, inverse, beta_bound,
                gamma_bound, pedestal, keep):
        _check(x, weight, bias, beta, gamma)
        stem = (not transposed) and weight.shape[1] < 4
        beta_e, gT = _gdn_operands_bf16(beta, gamma, beta_bound, gamma_bound, pedestal, kperm=True)
        kh, kw = weight.shape[2], weight.shape[3]
        epi = L.EPI_CONV_IGDN if inverse else L.EPI_CONV_GDN
        direct = False
        if stem:
            if x.requires_grad:
                raise NotImplementedError("the bf16 stem does not produce a gradient for the image")
            direct = bool(L.load().lic_stem_gdn_bf16_supported(weight.shape[1], weight.shape[0], kh, kw, stride, pad))
        if direct:
            # image -> normalised features in one launch, no column matrix (lic_stem_gdn_bf16); the backward pass
            # builds the columns its weight gradient needs from the saved image
            lib = L.load()
            src = _nhwc(x).float()
            B, Hi, Wi, Cin = src.shape
            Cout = weight.shape[0]
            Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, False)
            wp = prepared(weight, "bf16.stem16")
            if wp is None:
                wp = torch.empty((lib.lic_stem_weight_bf16_elems(Cout),), device=x.device, dtype=BF16)
                L.check(lib.lic_pack_stem_weight_bf16(_ptr(weight.contiguous()), _ptr(wp), Cout, _stream()),
                        "lic_pack_stem_weight_bf16")
            y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=BF16)
            conv_out = torch.empty_like(y) if keep else None
            norm = torch.empty_like(y) if keep and not norm_recomputed_bf16(Cout) else None
            from . import functional as F_
            if F_.KERNEL_TRACE is not None:
                F_.KERNEL_TRACE.add(f"stem_gdn_bf16_kernel<{Cout // 32}, {8 if Cout == 192 else 4}>")
            Pn = B * Ho * Wo
            F_._timed(f"stem_gdn_bf16_kernel<{Cout // 32}, {8 if Cout == 192 else 4}>",
                      2 * Pn * Cout * (kh * kw * Cin + Cout), 4 * src.numel() + 2 * Pn * Cout * (3 if keep else 1),
                      lambda: L.check(lib.lic_stem_gdn_bf16(_ptr(src), _ptr(wp), _ptr(bias), _ptr(gT), _ptr(beta_e), _ptr(y),
                                                            _ptr(conv_out), _ptr(norm), B, Hi, Wi, Cout, int(inverse),
                                                            _stream()), "lic_stem_gdn_bf16"))
        else:
            if stem:
                src, wp, (B, Ho, Wo, Cout, Cin, Kp, P) = _stem_columns_bf16(x, weight, stride, pad)
                geo = dict(B=1, Hi=1, Wi=P, Cin=Kp, Ho=1, Wo=P, Cout=Cout, kh=1, kw=1, stride=1, pad=0, transposed=False)
            else:
                src = _as_bf16_nhwc(x)
                B, Hi, Wi, Cin = src.shape
                Cout = weight.shape[1] if transposed else weight.shape[0]
                Ho, Wo = conv_out_size(Hi, Wi, kh, stride, pad, transposed, out_pad)
                wp = _pack_conv_weight_bf16(weight, transposed, False)
                geo = dict(B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, Cout=Cout, kh=kh, kw=kw, stride=stride, pad=pad,
                           transposed=transposed)
            y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=BF16)
            conv_out = torch.empty_like(y) if keep else None
            norm = torch.empty_like(y) if keep and not norm_recomputed_bf16(Cout) else None
            _igemm_bf16(src, wp, y, bias=bias, epilogue=epi, aux=gT, aux2=beta_e, out2=norm, out3=conv_out, **geo)
        ctx.cols = None
        if direct and keep and EARLY_STEM_COLUMNS:
            from . import functional as F_
            aux = F_.AUX_STREAM
            if aux is not None:
                # the image's column matrix (the operand of the stem's weight gradient) depends on the image alone: it is
                # built now, on the second stream beside the next layer's forward, instead of in the tail of the backward
                # pass where nothing else runs
                cur = torch.cuda.current_stream()
                aux.wait_stream(cur)
                with torch.cuda.stream(aux):
                    cols = _stem_columns_bf16(x, weight, stride, pad)[0]
                    ev = aux.record_event()
                cols.record_stream(cur)
                ctx.cols = (cols, ev)
        ctx.save_for_backward(src, weight, conv_out, norm, beta, gamma)
        ctx.cfg = (stride, pad, transposed, inverse, beta_bound, gamma_bound, pedestal, bias is not None, x.dtype, stem,
                   Cin, direct)
        return _nchw_view(y)

    @staticmethod
    def backward(ctx, gy):
        src, weight, conv_out, norm, beta, gamma = ctx.saved_tensors
        (stride, pad, transposed, inverse, beta_bound, gamma_bound, pedestal, has_bias, in_dtype, stem, Cin,
         direct) = ctx.cfg
        need = ctx.needs_input_grad
        want_db = has_bias and need[2]
        g_conv, dbeta, dgamma, db_pre = _gdn_backward_bf16(conv_out, norm, beta, gamma, _as_bf16_nhwc(gy), inverse,
                                                           beta_bound, gamma_bound, pedestal, True, need[3], need[4],
                                                           bias_from_dx=want_db)
        if direct and (need[1] or (has_bias and need[2])):   # src is the image: its columns for the weight gradient
            if ctx.cols is not None:     # built beside the forward pass
                src, ev = ctx.cols
                ctx.cols = None
                torch.cuda.current_stream().wait_event(ev)
            else:
                src = _stem_columns_bf16(_nchw_view(src), weight, stride, pad)[0]
        # (the bias gradient = column sums of g_conv: already there when it shared d-beta's launch pair)
        if stem:
            dx = None
            dw, db = _stem_backward_bf16(src, weight, g_conv, Cin, need[1], want_db and db_pre is None)
        else:
            dx, dw, db = _conv_backward_bf16(src, weight, g_conv, stride, pad, transposed, in_dtype, 0, need[0], need[1],
                                             want_db and db_pre is None)
        if db_pre is not None:
            db = db_pre
        return dx, dw, db, dbeta, dgamma, None, None, None, None, None, None, None, None, None


from .functional import will_backprop as _will_backprop  # noqa: E402


def fused_gdn_supported_bf16(cin: int, cout: int) -> bool:
    return bool(L.load().lic_igemm_bf16_fused_gdn_supported(int(cin), int(cout)))


def conv_gdn_bf16(x, weight, bias, beta, gamma, stride, padding, inverse, beta_bound, gamma_bound, pedestal=PEDESTAL,
                  transposed=False, output_padding=0):
    """`gdn_bf16(conv2d_bf16(x))` / `gdn_bf16(conv_transpose2d_bf16(x))` in one launch (see _ConvGDNBF16Fn)"""
    return _ConvGDNBF16Fn.apply(x, weight, bias, beta, gamma, stride, padding, output_padding, bool(transposed),
                                bool(inverse), float(beta_bound), float(gamma_bound), float(pedestal),
                                _will_backprop(x, weight, bias, beta, gamma))


def conv2d_bf16(x, weight, bias, stride, padding, out_f32=False, leaky=False, slope=0.01, tap_mask=0, out=None):
    """`out`: optional [B,Ho,Wo,Cout] channel slice of a wider contiguous bf16 NHWC buffer to write into"""
    return _ConvBF16Fn.apply(x, weight, bias, stride, padding, 0, False, out_f32, leaky, slope, tap_mask, out)


def conv_transpose2d_bf16(x, weight, bias, stride, padding, output_padding, out_f32=False, leaky=False, slope=0.01):
    return _ConvBF16Fn.apply(x, weight, bias, stride, padding, output_padding, True, out_f32, leaky, slope, 0)


def image_conv2d_bf16(x, weight, bias, stride, padding):
    return _ImageConvBF16Fn.apply(x, weight, bias, stride, padding)


def image_conv_transpose2d_bf16(x, weight, bias, stride, padding, output_padding):
    return _ImageConvTBF16Fn.apply(x, weight, bias, stride, padding, output_padding)


def gdn_bf16(x, beta, gamma, inverse, beta_bound, gamma_bound, pedestal=PEDESTAL):
    return _GDNBF16Fn.apply(x, beta, gamma, bool(inverse), float(beta_bound), float(gamma_bound), float(pedestal),
                            _will_backprop(x, beta, gamma))
