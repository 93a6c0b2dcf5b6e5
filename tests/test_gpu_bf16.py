"""bf16-storage mode (BASELINE config 3; not in the reference) on the GPU.

Declared tolerances (the 1e-4 bar of the north star applies to the fp32 path only):
  * kernels fed bf16-exact operands must match the fp32 oracle on the SAME rounded operands to
    fp32-accumulation accuracy (1e-4 of the tensor scale) when they write fp32, and to one bf16
    rounding (2^-8 relative, plus the tensor-scale floor) when they write bf16;
  * the whole model in bf16 mode vs the same model in fp32 mode: bpp / PSNR / loss within 3 %,
    every large parameter gradient with cosine similarity >= 0.98.
"""
import math

import numpy as np
import pytest
import torch

import golden_recipe as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    from neural_image_compression_amd import functional_bf16 as FB
    from oracle import oracle as O
    return nic, FB, O, torch.device("cuda:0")


def rb(a):
    """round a numpy fp32 array to bf16-representable values"""
    return torch.from_numpy(np.ascontiguousarray(a)).to(BF).float().numpy()


def dev(a, d, dtype=None, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(d)
    if t.dim() == 4:
        t = t.contiguous(memory_format=torch.channels_last)
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(grad)


def host(t):
    return t.detach().float().cpu().contiguous().numpy()


def scale_close(a, b, rel, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    s = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= rel * s, f"{what}: {np.abs(a - b).max():.3e} vs scale {s:.3e}"


def bf16_close(a, b, what):
    """one bf16 rounding of b: |a-b| <= 2^-8 |b| + 2^-8 * 1e-2 * scale"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    s = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b)
    assert (err <= 2.0 ** -8 * np.abs(b) + 2.0 ** -8 * 1e-2 * s + 1e-30).all(), f"{what}: {err.max():.3e} (scale {s:.3e})"


@pytest.mark.parametrize("k,s,p,ci,co,H,W,B,tr,op", [(5, 2, 2, 64, 64, 16, 16, 2, False, 0),
                                                     (5, 2, 2, 32, 192, 10, 18, 2, False, 0),
                                                     (3, 1, 1, 16, 24, 8, 8, 2, False, 0),
                                                     (5, 2, 2, 64, 64, 6, 6, 2, True, 1),
                                                     (5, 2, 2, 192, 32, 5, 3, 1, True, 1),
                                                     (1, 1, 0, 192, 80, 4, 6, 2, False, 0)])
@pytest.mark.parametrize("split", [0, 1, 3])   # K split across workgroups: automatic (these sizes split), never, 3 ways
def test_conv_bf16_ops(env, k, s, p, ci, co, H, W, B, tr, op, split):
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    F_.FORCE_IGEMM = (0, 0, split)
    try:
        _conv_bf16_ops(env, k, s, p, ci, co, H, W, B, tr, op)
    finally:
        F_.FORCE_IGEMM = None


@pytest.mark.parametrize("k,s,p,ci,co,H,W,B,tr,op", [(5, 2, 2, 64, 64, 40, 36, 2, False, 0),     # 720 rows: 2.8 tiles
                                                     (5, 2, 2, 128, 128, 34, 30, 1, False, 0),
                                                     (3, 1, 1, 72, 192, 20, 20, 1, False, 0),    # K tail, 192 columns
                                                     (5, 2, 2, 128, 64, 13, 11, 2, True, 1),     # 4 phases, ragged
                                                     (5, 2, 2, 64, 192, 12, 12, 1, True, 1),
                                                     (1, 1, 0, 192, 128, 17, 19, 1, False, 0)])
def test_conv_bf16_ops_eight_wave_tile(env, k, s, p, ci, co, H, W, B, tr, op):
    """the 256-row, 8-wave ping-pong variant (what the big layers dispatch) forced onto small ragged shapes: against
    the oracle, and bitwise against the 4-wave kernel (same chunk and k order per output)"""
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    names = set()
    F_.FORCE_IGEMM, F_.KERNEL_TRACE = (256, 0, 1), names
    try:
        _conv_bf16_ops(env, k, s, p, ci, co, H, W, B, tr, op)
        r = np.random.RandomState(ci + co)
        x = dev(rb(r.randn(B, ci, H, W).astype(np.float32)), d, BF)
        wshape = (ci, co, k, k) if tr else (co, ci, k, k)
        w = dev(rb((r.randn(*wshape) / math.sqrt(ci * k * k)).astype(np.float32)), d).contiguous()
        b = dev(r.randn(co).astype(np.float32), d)
        f = (lambda: FB.conv_transpose2d_bf16(x, w, b, s, p, op)) if tr else (lambda: FB.conv2d_bf16(x, w, b, s, p))
        with torch.no_grad():
            y8 = f()
            F_.FORCE_IGEMM = (128, 0, 1)
            y4 = f()
    finally:
        F_.FORCE_IGEMM, F_.KERNEL_TRACE = None, None
    assert any(n.startswith("igemm_bf16_kernel<256,") and n.endswith(", 8>") for n in names), names
    assert torch.equal(y8, y4)


def test_eight_wave_tile_at_full_size_beside_a_second_stream(env):
    """The 8-wave variant's groups hand chunks to each other through counted waits and barriers; a mistake there is
    a RACE that small quiet launches never lose (an early version passed everything above and produced NaNs only in
    the two-stream training step).  So: the real 128 -> 128, 5x5 s2 layer of config 3 (128 tiles per XCD) and its
    transposed twin, while a second stream keeps the GPU busy with other work, several times, bitwise against the
    4-wave kernel."""
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(32, 128, 128, 128, generator=g).to(d).contiguous(memory_format=torch.channels_last).to(BF)
    xs = torch.randn(32, 128, 64, 64, generator=g).to(d).contiguous(memory_format=torch.channels_last).to(BF)
    w = (torch.randn(128, 128, 5, 5, generator=g) / 56.0).to(d)
    b = torch.randn(128, generator=g).to(d)
    a = torch.randn(4096, 4096, device=d, dtype=BF)
    side = torch.cuda.Stream()
    fns = [lambda: FB.conv2d_bf16(x, w, b, 2, 2), lambda: FB.conv_transpose2d_bf16(xs, w, b, 2, 2, 1)]
    with torch.no_grad():
        F_.FORCE_IGEMM = (128, 0, 1)
        try:
            ref = [f() for f in fns]
            torch.cuda.synchronize()
            F_.FORCE_IGEMM = (256, 0, 1)
            for it in range(6):
                with torch.cuda.stream(side):
                    for _ in range(6):
                        a @ a
                outs = [f() for f in fns]
                torch.cuda.synchronize()
                for o, r_ in zip(outs, ref):
                    assert torch.equal(o, r_), f"iteration {it}: the 8-wave tile differs from the 4-wave kernel"
        finally:
            F_.FORCE_IGEMM = None


@pytest.mark.parametrize("ci,co,H,W,B", [(64, 128, 37, 45, 2),     # partial tiles in both directions, 2 chunks
                                          (128, 128, 20, 70, 1),    # two column tiles, the second 3 pixels wide
                                          (192, 128, 16, 16, 3),    # 6 chunks (three trips of the two-chunk body)
                                          (64, 128, 5, 3, 2)])      # a tile that is almost all padding
def test_halo_conv_forced_onto_small_ragged_shapes(env, ci, co, H, W, B):
    """the halo-resident 5x5 stride-2 kernel (lic_halo_bf16.h; what the 128-channel stride-2 layers of config 3
    dispatch at full size) forced onto small ragged shapes, forward and backward against the oracle"""
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    names = set()
    F_.FORCE_IGEMM, F_.KERNEL_TRACE = (512, 0, 1), names
    try:
        _conv_bf16_ops(env, 5, 2, 2, ci, co, H, W, B, False, 0)
    finally:
        F_.FORCE_IGEMM, F_.KERNEL_TRACE = None, None
    assert "halo_conv_bf16_kernel<2, false, 0>" in names, names


@pytest.mark.parametrize("ci,co,H,W,B", [(128, 128, 19, 37, 2),    # partial q tiles in both directions
                                          (64, 128, 8, 8, 3),       # 2 chunks: one trip of every phase body
                                          (192, 128, 5, 40, 1),     # 6 chunks, two column tiles
                                          (128, 128, 1, 1, 2)])     # a tile that is almost all padding
def test_halo_transposed_conv_forced_onto_small_ragged_shapes(env, ci, co, H, W, B):
    """the transposed halo kernel (lic_halot_bf16.h: four phases on one resident input patch) forced onto small ragged
    shapes, forward and backward against the oracle (with ci = 128 the data gradient runs the strided halo kernel)"""
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    names = set()
    F_.FORCE_IGEMM, F_.KERNEL_TRACE = (512, 0, 1), names
    try:
        _conv_bf16_ops(env, 5, 2, 2, ci, co, H, W, B, True, 1)
    finally:
        F_.FORCE_IGEMM, F_.KERNEL_TRACE = None, None
    assert "halo_convt_bf16_kernel<2, false>" in names, names


def test_halo_conv_at_full_size_beside_a_second_stream(env):
    """The halo kernel hands LDS buffers from the DMA to the fragment reads with counted `vmcnt` waits and ONE barrier
    per channel chunk, and keeps weight fragments in flight in registers: a mistake there is a race that small quiet
    launches never lose.  The real 128 -> 128 stride-2 layer of config 3 at batch 32 (two tiles per persistent
    workgroup: the cross-tile prefetch runs), picked by the AUTOMATIC dispatch, beside a busy second stream, four
    rounds: equal to the implicit-GEMM tile within fp32 summation order (the K order differs: chunk-major), and
    bit-identical from launch to launch; also with the fused LeakyReLU and an fp32 output."""
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(32, 128, 128, 128, generator=g).to(d).contiguous(memory_format=torch.channels_last).to(BF)
    xs = torch.randn(32, 128, 64, 64, generator=g).to(d).contiguous(memory_format=torch.channels_last).to(BF)
    w = (torch.randn(128, 128, 5, 5, generator=g) / 56.0).to(d)
    b = torch.randn(128, generator=g).to(d)
    a = torch.randn(4096, 4096, device=d, dtype=BF)
    side = torch.cuda.Stream()
    fns = [lambda: FB.conv2d_bf16(x, w, b, 2, 2), lambda: FB.conv2d_bf16(x, w, b, 2, 2, leaky=True, slope=0.01),
           lambda: FB.conv2d_bf16(x, w, b, 2, 2, out_f32=True),
           lambda: FB.conv_transpose2d_bf16(xs, w, b, 2, 2, 1),          # the transposed halo kernel, 512 tiles
           lambda: FB.conv_transpose2d_bf16(xs, w, b, 2, 2, 1, out_f32=True)]
    names = set()
    with torch.no_grad():
        F_.FORCE_IGEMM = (128, 0, 1)
        try:
            ref = [f().float() for f in fns]
            torch.cuda.synchronize()
            F_.FORCE_IGEMM, F_.KERNEL_TRACE = None, names
            first = None
            for it in range(4):
                with torch.cuda.stream(side):
                    for _ in range(6):
                        a @ a
                outs = [f() for f in fns]
                torch.cuda.synchronize()
                for o, r_ in zip(outs, ref):
                    err = (o.float() - r_).abs()
                    tol = 2.0 ** -7 * r_.abs() + 2.0 ** -7 * 1e-2 * r_.abs().max()   # one bf16 ulp either way
                    assert bool((err <= tol).all()), f"iteration {it}: {float(err.max()):.3e}"
                if first is None:
                    first = outs
                else:
                    for o, f0 in zip(outs, first):
                        assert torch.equal(o, f0), f"iteration {it}: the halo kernel is not bit-repeatable"
        finally:
            F_.FORCE_IGEMM, F_.KERNEL_TRACE = None, None
    assert names == {"halo_conv_bf16_kernel<2, false, 0>", "halo_convt_bf16_kernel<2, false>"}, names


def test_dispatched_eight_wave_tiles_at_full_size_beside_a_second_stream(env):
    """What the 192-channel layers ACTUALLY dispatch (VERDICT r2 / ADVICE r2): the 8-wave ping-pong tile is automatic
    only for TN = 3 (BPASS = 2: the weight panel's pieces 12..15 wrap onto 0..3 and are DMA'd a second time half a
    chunk later by the other wave group), plain and with the fused pool -- the test above forces the tile onto the
    128-channel layer, which takes neither path.  The real 192 -> 192 5x5 s2 layer of config 2h, its transposed twin
    and the conv + GDN launch with the AUTOMATIC tile (the traced names must be the 8-wave TN = 3 instantiations),
    beside a busy second stream, four rounds, bitwise against the 4-wave kernel."""
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    from neural_image_compression_amd.layers import GDN
    g = torch.Generator(device="cpu").manual_seed(21)
    C = 192
    x = torch.randn(32, C, 128, 128, generator=g).to(d).contiguous(memory_format=torch.channels_last).to(BF)
    xs = torch.randn(32, C, 64, 64, generator=g).to(d).contiguous(memory_format=torch.channels_last).to(BF)
    w = (torch.randn(C, C, 5, 5, generator=g) / 69.0).to(d)
    b = torch.randn(C, generator=g).to(d)
    gd = GDN(C).to(d)
    gargs = (gd.beta, gd.gamma, 2, 2, False, gd.beta_reparam.bound_value, gd.gamma_reparam.bound_value,
             gd.beta_reparam.pedestal_value)
    a = torch.randn(4096, 4096, device=d, dtype=BF)
    side = torch.cuda.Stream()
    fns = [lambda: FB.conv2d_bf16(x, w, b, 2, 2), lambda: FB.conv_transpose2d_bf16(xs, w, b, 2, 2, 1),
           lambda: FB.conv_gdn_bf16(x, w, b, *gargs)]
    names = set()
    with torch.no_grad():
        F_.FORCE_IGEMM = (128, 0, 1)
        try:
            ref = [f() for f in fns]
            torch.cuda.synchronize()
            F_.FORCE_IGEMM, F_.KERNEL_TRACE = None, names
            for it in range(4):
                with torch.cuda.stream(side):
                    for _ in range(6):
                        a @ a
                outs = [f() for f in fns]
                torch.cuda.synchronize()
                for k, (o, r_) in enumerate(zip(outs, ref)):
                    assert torch.equal(o, r_), f"iteration {it}, launch {k}: the 8-wave tile differs from the 4-wave kernel"
        finally:
            F_.FORCE_IGEMM, F_.KERNEL_TRACE = None, None
    assert "igemm_bf16_kernel<256, 3, false, false, 4, 8>" in names, names
    assert "igemm_bf16_kernel<256, 3, false, true, 4, 8>" in names, names


@pytest.mark.parametrize("M,K", [(128, 3), (192, 1)])
def test_two_stream_bf16_steps_are_bitwise_independent_of_the_tile_switches(env, M, K):
    """config 3 / config 2h training steps (two HIP streams) with the 8-wave variant off (LIC_BF16_PP=0), with the
    three-buffer ring (LIC_BF16_RING=3) and with the defaults: every variant sums an output in the same chunk and k
    order, so the losses after two optimizer steps must be BIT-identical -- a hand-over race in any of them shows up
    as a difference (the one of round 2 produced NaNs exactly here)."""
    nic, FB, O, d = env
    import os

    def run():
        torch.manual_seed(0)
        model = nic.JointAutoregressiveHierarchical(M, K).to(d)
        model.set_precision("bf16")
        model.overlap_branches = True
        opt = nic.FusedAdam(model.parameters(), lr=1e-4)
        g = torch.Generator(device="cpu").manual_seed(1234)
        x = torch.rand(32, 3, 256, 256, generator=g).to(d).contiguous(memory_format=torch.channels_last)
        gn = torch.Generator(device="cpu").manual_seed(4321)
        uz = torch.rand(32, M, 4, 4, generator=gn).to(d)
        uy = torch.rand(32, M, 16, 16, generator=gn).to(d)
        losses = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            res = nic.rd_loss(model(x, noise=(uz, uy)), x, 0.01, sync=False)
            res["loss"].backward()
            opt.step()
            losses.append(float(res["loss"].detach()))
        torch.cuda.synchronize()
        return losses

    saved = {k: os.environ.get(k) for k in ("LIC_BF16_PP", "LIC_BF16_RING")}
    try:
        base = run()
        assert all(math.isfinite(v) for v in base), base
        for k, v in (("LIC_BF16_PP", "0"), ("LIC_BF16_RING", "3")):
            os.environ[k] = v
            got = run()
            del os.environ[k]
            assert got == base, (k, got, base)
        assert run() == base
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_conv_bf16_split_is_batch_invariant(env):
    """the K split is chosen from per-image geometry: an image's output bits do not depend on its batch"""
    nic, FB, O, d = env
    r = np.random.RandomState(3)
    x = dev(rb(r.randn(6, 128, 8, 8).astype(np.float32)), d, BF)
    w = dev(rb((r.randn(128, 128, 5, 5) / 56.0).astype(np.float32)), d).contiguous()
    b = dev(r.randn(128).astype(np.float32), d)
    with torch.no_grad():
        y6 = FB.conv2d_bf16(x, w, b, 2, 2)
        y2 = FB.conv2d_bf16(x[2:4].contiguous(memory_format=torch.channels_last), w, b, 2, 2)
    assert torch.equal(y6[2:4], y2)
    # a 192-channel layer at a batch large enough for the 8-wave tile (>= 256 tiles of 256 rows): the tile
    # choice must not switch the K split of this tiny-spatial layer off
    xb = dev(rb(r.randn(4096, 192, 8, 8).astype(np.float32)), d, BF)
    wb = dev(rb((r.randn(192, 192, 5, 5) / 69.0).astype(np.float32)), d).contiguous()
    bb = dev(r.randn(192).astype(np.float32), d)
    with torch.no_grad():
        big = FB.conv2d_bf16(xb, wb, bb, 2, 2)
        small = FB.conv2d_bf16(xb[5:7].contiguous(memory_format=torch.channels_last), wb, bb, 2, 2)
    assert torch.equal(big[5:7], small)
    # the halo-resident variant (chunk-major K order) is chosen from per-image geometry too: 64x64 inputs of the
    # 128-channel layer take it at every batch
    xh = dev(rb(r.randn(5, 128, 64, 64).astype(np.float32)), d, BF)
    with torch.no_grad():
        y5 = FB.conv2d_bf16(xh, w, b, 2, 2)
        y1 = FB.conv2d_bf16(xh[3:4].contiguous(memory_format=torch.channels_last), w, b, 2, 2)
    assert torch.equal(y5[3:4], y1)


def _conv_bf16_ops(env, k, s, p, ci, co, H, W, B, tr, op):
    nic, FB, O, d = env
    r = np.random.RandomState(ci + co)
    x = rb(r.randn(B, ci, H, W).astype(np.float32))
    wshape = (ci, co, k, k) if tr else (co, ci, k, k)
    w = rb((r.randn(*wshape) / math.sqrt(ci * k * k)).astype(np.float32))
    b = r.randn(co).astype(np.float32)
    tw, tb = dev(w, d).contiguous().requires_grad_(True), dev(b, d, grad=True)
    fwd = (lambda t, f32: FB.conv_transpose2d_bf16(t, tw, tb, s, p, op, f32)) if tr else \
        (lambda t, f32: FB.conv2d_bf16(t, tw, tb, s, p, f32))
    y_ref = O.convT2d_fwd(x, w, b, s, p, op) if tr else O.conv2d_fwd(x, w, b, s, p)
    # fp32 output: only the accumulation order differs from the oracle
    tx = dev(x, d, BF, grad=True)
    y32 = fwd(tx, True)
    assert y32.dtype == torch.float32
    scale_close(host(y32), y_ref, 1e-4, "y (fp32 out)")
    dy = rb(r.randn(*y_ref.shape).astype(np.float32))
    y32.backward(dev(dy, d))
    dx, dw, db = (O.convT2d_bwd(x, w, dy, s, p, op) if tr else O.conv2d_bwd(x, w, dy, s, p))
    bf16_close(host(tx.grad), dx, "dx (bf16 out)")
    scale_close(host(tw.grad), dw, 1e-4, "dw")
    scale_close(host(tb.grad), db, 1e-4, "db")
    # bf16 output: one extra rounding
    ybf = fwd(dev(x, d, BF), False)
    assert ybf.dtype == BF
    bf16_close(host(ybf), y_ref, "y (bf16 out)")


@pytest.mark.parametrize("P,Cc", [(1000, 72), (37, 640), (8192, 128), (1, 8)])
@pytest.mark.parametrize("deferred", [False, True])
def test_leaky_backward_with_bias_sum_in_one_pass(env, P, Cc, deferred):
    """lic_leaky_bwd_colsum_bf16 (the gradient of a conv -> LeakyReLU layer, Components.py:69-73 / ParametersModels.py:22-34,
    masked and column-summed in one pass) = lic_leaky_bwd_bf16 followed by lic_colsum_bf16, bit for bit; `deferred`: the
    sums' second stage through lic_reduce_batch"""
    nic, FB, O, d = env
    import ctypes as C
    from neural_image_compression_amd import _lib as L
    lib = L.load()
    g = torch.Generator(device="cpu").manual_seed(P + Cc)
    y = torch.randn(P, Cc, generator=g).to(d).to(BF)
    dy = torch.randn(P, Cc, generator=g).to(d).to(BF)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    nbytes = lib.lic_colsum_bf16_workspace_bytes(P, Cc)
    # two passes
    dx_ref = torch.empty_like(y)
    L.check(lib.lic_leaky_bwd_bf16(ptr(y), ptr(dy), ptr(dx_ref), y.numel(), 0.01, st), "leaky")
    ws = torch.empty((nbytes + 3) // 4, device=d)
    db_ref = torch.empty(Cc, device=d)
    L.check(lib.lic_colsum_bf16(ptr(dx_ref), Cc, P, Cc, 1.0, ptr(db_ref), ptr(ws), nbytes, st), "colsum")
    # one pass
    dx = torch.full_like(y, float("nan"))
    ws2 = torch.empty((nbytes + 3) // 4, device=d)
    db = torch.full((Cc,), float("nan"), device=d)
    job = L.ReduceJob()
    L.check(lib.lic_leaky_bwd_colsum_bf16(ptr(y), ptr(dy), ptr(dx), P, Cc, 0.01, ptr(db), ptr(ws2), nbytes,
                                          C.byref(job) if deferred else None, st), "fused")
    if deferred:
        L.check(lib.lic_reduce_batch(C.byref(job), 1, st), "reduce")
    torch.cuda.synchronize()
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16))
    assert torch.equal(db.view(torch.int32), db_ref.view(torch.int32))
    # and against plain arithmetic on the same bf16 values
    ref = torch.where(y.float() > 0, dy.float(), (dy.float() * 0.01).to(BF).float())
    assert torch.equal(dx.float(), ref)
    assert torch.allclose(db.cpu().double(), ref.double().sum(0).cpu(), rtol=1e-5, atol=1e-4 * max(1.0, P ** 0.5))


def test_image_layers_bf16(env):
    nic, FB, O, d = env
    r = np.random.RandomState(3)
    x = r.rand(2, 3, 32, 32).astype(np.float32)
    w = rb((r.randn(64, 3, 5, 5) / math.sqrt(75)).astype(np.float32))
    b = r.randn(64).astype(np.float32)
    tw, tb = dev(w, d).contiguous().requires_grad_(True), dev(b, d, grad=True)
    y = FB.image_conv2d_bf16(dev(x, d), tw, tb, 2, 2)
    y_ref = O.conv2d_fwd(rb(x), w, b, 2, 2)
    bf16_close(host(y), y_ref, "stem y")
    dy = rb(r.randn(*y_ref.shape).astype(np.float32))
    y.backward(dev(dy, d, BF))
    _, dw, db = O.conv2d_bwd(rb(x), w, dy, 2, 2, need_dx=False)
    scale_close(host(tw.grad), dw, 1e-4, "stem dw")
    scale_close(host(tb.grad), db, 1e-4, "stem db")
    # head
    xh = rb(r.randn(2, 64, 8, 8).astype(np.float32))
    wt = rb((r.randn(64, 3, 5, 5) / math.sqrt(64 * 25)).astype(np.float32))
    bt = r.randn(3).astype(np.float32)
    twt, tbt = dev(wt, d).contiguous().requires_grad_(True), dev(bt, d, grad=True)
    txh = dev(xh, d, BF, grad=True)
    out = FB.image_conv_transpose2d_bf16(txh, twt, tbt, 2, 2, 1)
    assert out.dtype == torch.float32
    out_ref = O.convT2d_fwd(xh, wt, bt, 2, 2, 1)
    # the per-tap columns are bf16: up to 9 rounded terms per output
    scale_close(host(out), out_ref, 2e-2, "head out")
    g = r.randn(*out_ref.shape).astype(np.float32)
    out.backward(dev(g, d))
    dx, dw, db = O.convT2d_bwd(xh, wt, rb(g), 2, 2, 1)
    scale_close(host(txh.grad), dx, 2e-2, "head dx")
    scale_close(host(twt.grad), dw, 1e-4, "head dw")
    dbf = O.convT2d_bwd(xh, wt, g, 2, 2, 1)[2]
    scale_close(host(tbt.grad), dbf, 1e-4, "head db")


@pytest.mark.parametrize("B,H,W", [(2, 21, 19), (1, 64, 64), (3, 6, 5), (1, 2, 2)])
def test_im2col_rgb5_matches_the_generic_gather(env, monkeypatch, B, H, W):
    """the one-lane-per-pixel im2col of the RGB 5x5 stride-2 geometry (lic_im2col_bf16's fast path) against the generic
    table-driven gather it replaces: every border case, bit for bit"""
    nic, FB, O, d = env
    from neural_image_compression_amd import _lib as L
    from neural_image_compression_amd.functional import _ptr, _stream
    lib = L.load()
    x = torch.randn(B, H, W, 3, device=d)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    cols = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("LIC_IM2COL_GENERIC", "1")
        col = torch.full((B * Ho * Wo, 80), float("nan"), device=d, dtype=BF)
        L.check(lib.lic_im2col_bf16(_ptr(x), _ptr(col), B, H, W, 3, Ho, Wo, 5, 5, 2, 2, 80, _stream()), "lic_im2col_bf16")
        torch.cuda.synchronize()
        cols.append(col)
    assert torch.equal(cols[0].view(torch.int16), cols[1].view(torch.int16))
    # against the definition, for one sample of entries
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2))
    ref = xp.unfold(2, 5, 2).unfold(3, 5, 2)            # [B, 3, Ho, Wo, 5, 5]
    ref = ref.permute(0, 2, 3, 4, 5, 1).reshape(B * Ho * Wo, 75).to(BF)
    assert torch.equal(cols[0][:, :75].view(torch.int16), ref.contiguous().view(torch.int16))
    assert float(cols[0][:, 75:].abs().max()) == 0.0


# the RGB head without column matrices (lic_head_convt_bf16: features -> image in one launch; lic_stem_conv_bf16: its data
# gradient as a direct convolution of the image gradient): partial tiles in both directions, every supported width,
# against the oracle on bf16-exact operands (fp32 accumulation of exact products: tight) and against the column-matrix
# route (which rounds the per-tap columns to bf16: its own tolerance)
@pytest.mark.parametrize("C,B,Hi,Wi", [(64, 2, 5, 7), (128, 2, 9, 40), (128, 1, 8, 64), (192, 1, 6, 33), (128, 3, 1, 1)])
def test_head_direct_bf16(env, monkeypatch, C, B, Hi, Wi):
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    r = np.random.RandomState(C + Hi + Wi)
    xh = rb(r.randn(B, C, Hi, Wi).astype(np.float32))
    wt = rb((r.randn(C, 3, 5, 5) / math.sqrt(C * 25)).astype(np.float32))
    bt = r.randn(3).astype(np.float32)
    g = r.randn(B, 3, 2 * Hi, 2 * Wi).astype(np.float32)

    def run():
        twt, tbt = dev(wt, d).contiguous().requires_grad_(True), dev(bt, d, grad=True)
        txh = dev(xh, d, BF, grad=True)
        F_.KERNEL_TRACE = set()
        out = FB.image_conv_transpose2d_bf16(txh, twt, tbt, 2, 2, 1)
        out.backward(dev(g, d))
        names, F_.KERNEL_TRACE = F_.KERNEL_TRACE, None
        return host(out), host(txh.grad), host(twt.grad), host(tbt.grad), names

    out, dx, dw, db, names = run()
    assert any("head_convt_bf16_kernel" in n for n in names) and any("plain" in n for n in names), names
    out_ref = O.convT2d_fwd(xh, wt, bt, 2, 2, 1)
    scale_close(out, out_ref, 2e-5, "head out (direct)")
    dx_ref, dw_ref, _ = O.convT2d_bwd(xh, wt, rb(g), 2, 2, 1)
    bf16_close(dx, dx_ref, "head dx (direct)")
    scale_close(dw, dw_ref, 1e-4, "head dw")
    scale_close(db, O.convT2d_bwd(xh, wt, g, 2, 2, 1)[2], 1e-4, "head db")
    monkeypatch.setenv("LIC_BF16_HEAD_DIRECT", "0")
    out2, dx2, dw2, db2, names2 = run()
    assert not any("head_convt_bf16_kernel" in n for n in names2), names2
    scale_close(out, out2, 2e-2, "direct vs column-matrix route: out")
    scale_close(dx, dx2, 2e-2, "direct vs column-matrix route: dx")
    assert np.array_equal(dw, dw2) and np.array_equal(db, db2)   # (the weight gradient still runs from the columns)


@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_bf16(env, inverse):
    nic, FB, O, d = env
    from neural_image_compression_amd.layers import GDN
    C = 64
    r = np.random.RandomState(5)
    x = rb(r.randn(2, C, 6, 5).astype(np.float32))
    m = GDN(C, inverse=inverse).to(d)
    beta_p, gamma_p = R.make_param("g.beta", (C,), 3), R.make_param("g.gamma", (C, C), 3)
    with torch.no_grad():
        m.beta.copy_(torch.from_numpy(beta_p))
        m.gamma.copy_(torch.from_numpy(gamma_p))
    tx = dev(x, d, BF, grad=True)
    y = m(tx, bf16=True)
    beta_e, gamma_e = O.gdn_reparam(beta_p, 1e-6), O.gdn_reparam(gamma_p, 0.0)
    y_ref, nrm = O.gdn_fwd(x, beta_e, gamma_e, inverse)
    scale_close(host(y), y_ref, 2e-2, "gdn y")  # x^2 and gamma are rounded to bf16 before the contraction
    dy = rb(r.randn(*x.shape).astype(np.float32))
    y.backward(dev(dy, d, BF))
    dx, dbe, dge = O.gdn_bwd(x, nrm, gamma_e, dy, inverse)
    scale_close(host(tx.grad), dx, 3e-2, "gdn dx")
    scale_close(host(m.beta.grad), O.gdn_reparam_bwd(beta_p, dbe, 1e-6), 3e-2, "dbeta")
    scale_close(host(m.gamma.grad), O.gdn_reparam_bwd(gamma_p, dge, 0.0), 3e-2, "dgamma")


# the one-sweep GDN / IGDN backward (lic_gdn_bwd_bf16: t and dx from one read of g, x, norm) against the oracle on
# small ragged cases and against the two-launch route (lic_gdn_dnorm_bf16 + the GDN_BWD epilogue of lic_igemm_bf16)
# up to more tiles than the persistent grid has workgroups.  t is bitwise the two-launch route's; dx sums each group
# of 16 channels in another order: fp32 rounding of the pool, at most the odd bf16 flip of the stored result
@pytest.mark.parametrize("C,P,inverse", [(64, 1, False), (64, 37, True), (128, 131, False), (128, 1000, True),
                                         (64, 128 * 2048 + 77, False), (128, 128 * 700 + 5, True)])
def test_gdn_bwd_one_sweep_bf16(env, C, P, inverse):
    nic, FB, O, d = env
    from neural_image_compression_amd import _lib as L
    from neural_image_compression_amd.functional import _ptr, _stream
    lib = L.load()
    assert lib.lic_gdn_bwd_bf16_supported(C) == 1 and lib.lic_gdn_bwd_bf16_supported(192) == 0
    r = np.random.RandomState(C + P)
    gen = torch.Generator(device="cpu").manual_seed(C + P)
    x = torch.randn(P, C, generator=gen).to(BF).to(d)
    g = torch.randn(P, C, generator=gen).to(BF).to(d)
    nrm = (torch.rand(P, C, generator=gen) * 3.0 + 0.25).to(BF).to(d)
    gamma_e = torch.from_numpy(rb(np.abs(r.randn(C, C)).astype(np.float32) * 0.05)).to(d)   # [norm index][x index]
    gp1 = FB._pack_bf16(gamma_e, 1, C, C, 0, C, 1, kperm=True)
    dx1, t1 = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.lic_gdn_bwd_bf16(_ptr(g), _ptr(x), _ptr(nrm), _ptr(gp1), _ptr(dx1), _ptr(t1), None, None, P, C, int(inverse),
                                 _stream()), "lic_gdn_bwd_bf16")
    # two-launch route
    t2, dx2 = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.lic_gdn_dnorm_bf16(_ptr(g), _ptr(x), _ptr(nrm), _ptr(t2), x.numel(), int(inverse), _stream()), "dnorm")
    gp2 = FB._pack_bf16(gamma_e, 1, C, C, 0, C, 1)
    FB._igemm_bf16(t2, gp2, dx2, B=1, Hi=1, Wi=P, Cin=C, Ho=1, Wo=P, Cout=C, kh=1, kw=1, stride=1, pad=0,
                   transposed=False, epilogue=L.EPI_IGDN_BWD if inverse else L.EPI_GDN_BWD, aux=g, aux2=x, aux3=nrm)
    torch.cuda.synchronize()
    assert torch.equal(t1, t2), "t differs from lic_gdn_dnorm_bf16"
    a, b = dx1.float(), dx2.float()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2.0 ** -7 * scale, (float((a - b).abs().max()), scale)
    assert float((a != b).float().mean()) < 0.02     # the odd last-bit flip, no more
    if P <= 1000:
        # the oracle on the same bf16 inputs (x as [1, C, P, 1] NCHW)
        xn = x.float().cpu().numpy().T.reshape(1, C, P, 1)
        gn = g.float().cpu().numpy().T.reshape(1, C, P, 1)
        nn = nrm.float().cpu().numpy().T.reshape(1, C, P, 1)
        dxo = O.gdn_bwd(xn, nn, gamma_e.cpu().numpy(), gn, inverse)[0]
        scale_close(a.cpu().numpy().T.reshape(1, C, P, 1), dxo, 2e-2, "one-sweep dx against the oracle")
    # with the per-workgroup column sums: same t and dx bits, and the rows add up to the column sums of the bf16 tensors
    rows = lib.lic_gdn_bwd_bf16_partial_rows(P)
    pt = torch.full((rows, C), float("nan"), device=d)
    pdx = torch.full((rows, C), float("nan"), device=d)
    dx3, t3 = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.lic_gdn_bwd_bf16(_ptr(g), _ptr(x), _ptr(nrm), _ptr(gp1), _ptr(dx3), _ptr(t3), _ptr(pt), _ptr(pdx), P, C,
                                 int(inverse), _stream()), "lic_gdn_bwd_bf16 (column sums)")
    torch.cuda.synchronize()
    assert torch.equal(t3, t1) and torch.equal(dx3, dx1)
    for part, src, what in ((pt, t1, "t"), (pdx, dx1, "dx")):
        ref = src.double().sum(0)
        got = part.double().sum(0)
        tol = 1e-5 * float(src.double().abs().sum(0).max()) + 1e-6     # fp32 partial sums of bf16 values
        assert float((got - ref).abs().max()) <= tol, (what, float((got - ref).abs().max()), tol)
    assert lib.lic_gdn_bwd_bf16(_ptr(g), _ptr(x), _ptr(nrm), _ptr(gp1), _ptr(dx3), _ptr(t3), _ptr(pt), None, P, C, 0,
                                _stream()) == -1   # both or neither
    # the recomputing variant (no stored norm: norm = beta + x^2 . gamma^T formed in the kernel as the forward pass forms
    # it) against the reading variant fed with that norm: t and dx agree up to the odd bf16 flip of the recomputed norm
    beta_e = (torch.rand(C, generator=gen) * 0.5 + 0.1).to(d)
    sq = (x.float() ** 2).to(BF).float()
    nrm_re = (beta_e[None, :] + sq @ gamma_e.to(BF).float().t()).to(BF)
    gpT = FB._pack_bf16(gamma_e, 1, C, C, 0, 1, C, kperm=True)
    dx4, t4 = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.lic_gdn_bwd_bf16(_ptr(g), _ptr(x), _ptr(nrm_re), _ptr(gp1), _ptr(dx4), _ptr(t4), None, None, P, C,
                                 int(inverse), _stream()), "lic_gdn_bwd_bf16")
    dx5, t5 = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.lic_gdn_bwd_bf16_recompute(_ptr(g), _ptr(x), _ptr(gp1), _ptr(gpT), _ptr(beta_e), _ptr(dx5), _ptr(t5), _ptr(pt),
                                           _ptr(pdx), P, C, int(inverse), _stream()), "lic_gdn_bwd_bf16_recompute")
    torch.cuda.synchronize()
    for a5, a4, what in ((t5, t4, "t"), (dx5, dx4, "dx")):
        a5, a4 = a5.float(), a4.float()
        sc = float(a4.abs().max())
        assert float((a5 - a4).abs().max()) <= 2.0 ** -5 * sc, (what, float((a5 - a4).abs().max()), sc)
        assert float(((a5 - a4).abs() > 2.0 ** -9 * sc).float().mean()) < 0.02, what
    # unsupported widths and misaligned pointers are refused, not mis-run
    assert lib.lic_gdn_bwd_bf16(_ptr(g), _ptr(x), _ptr(nrm), _ptr(gp1), _ptr(dx1), _ptr(t1), None, None, P, 192, 0, _stream()) \
        == -2   # LIC_ERR_UNSUPPORTED


# conv -> GDN in one launch (LIC_EPI_CONV_GDN of lic_igemm_bf16): every N tile, ragged M, the 4-phase transposed
# gather and the RGB stem, against the two-launch path.  Same rounding points (x and x^2 to bf16, fp32 norm); the
# fused pool sums each group of 16 channels in another order, so: y within one bf16 rounding, gradients (which see
# the occasional 1-ulp flip of the stored bf16 norm) within 1 % of their scale -- and, through the two-launch path,
# the oracle tolerances of the tests above
@pytest.mark.parametrize("k,s,p,ci,co,H,W,B,tr,op,inverse,bm", [
    (5, 2, 2, 64, 64, 11, 9, 2, False, 0, False, 64),     # 64-column tile, ragged rows
    (5, 2, 2, 64, 128, 16, 16, 2, False, 0, False, 128),  # 128x128 tile
    (5, 2, 2, 128, 128, 9, 7, 3, False, 0, False, 64),
    (3, 1, 1, 72, 192, 8, 8, 2, False, 0, False, 128),    # 192-column tile, K tail (72 = 2 chunks + 8)
    (5, 2, 2, 128, 192, 7, 5, 2, True, 1, True, 64),      # transposed, 4 phases, IGDN
    (5, 2, 2, 64, 128, 8, 8, 2, True, 1, True, 128),
    (5, 2, 2, 3, 128, 20, 18, 2, False, 0, False, 128),   # RGB stem: image -> features in one launch (lic_stem_gdn_bf16)
    (5, 2, 2, 3, 192, 12, 12, 1, False, 0, False, 64),
    (5, 2, 2, 3, 64, 21, 19, 3, False, 0, False, 64),     # odd sizes: every border case of the window loads
    (5, 2, 2, 3, 128, 96, 80, 3, False, 0, False, 128),   # 45 tiles: the persistent loop
    (3, 2, 1, 3, 128, 12, 12, 2, False, 0, False, 128),   # a stem the direct kernel does not cover: columns + GEMM
    (5, 2, 2, 64, 128, 40, 36, 2, False, 0, False, 256),  # the 8-wave ping-pong tile with the fused pool
    (5, 2, 2, 128, 192, 18, 22, 1, False, 0, False, 256),
    (5, 2, 2, 64, 64, 13, 11, 2, True, 1, True, 256),     # ... transposed, 4 phases, IGDN
    (5, 2, 2, 64, 128, 37, 45, 2, False, 0, False, 512),  # the halo-resident kernel with the fused pool: partial tiles
    (5, 2, 2, 128, 128, 64, 64, 3, False, 0, False, 512), # ... four full tiles per image, two per workgroup and more
    (5, 2, 2, 192, 128, 16, 20, 2, False, 0, True, 512),  # ... six chunks, IGDN
    (5, 2, 2, 64, 128, 19, 21, 2, True, 1, True, 512),    # the transposed halo kernel with the fused pool (IGDN)
    (5, 2, 2, 128, 128, 32, 32, 2, True, 1, False, 512),  # ... four full tiles per image, GDN
])
def test_conv_gdn_fused_bf16_matches_two_launches(env, k, s, p, ci, co, H, W, B, tr, op, inverse, bm):
    nic, FB, O, d = env
    from neural_image_compression_amd import functional as F_
    from neural_image_compression_amd.layers import GDN
    r = np.random.RandomState(11)
    x = rb(r.randn(B, ci, H, W).astype(np.float32))
    wshape = (ci, co, k, k) if tr else (co, ci, k, k)
    w = torch.from_numpy(rb(r.randn(*wshape).astype(np.float32) / math.sqrt(ci * k * k))).to(d).requires_grad_(True)
    b = torch.from_numpy(rb(0.1 * r.randn(co).astype(np.float32))).to(d).requires_grad_(True)
    g = GDN(co, inverse=inverse).to(d)
    with torch.no_grad():
        g.beta.copy_(torch.from_numpy(R.make_param("g.beta", (co,), 3)))
        g.gamma.copy_(torch.from_numpy(R.make_param("g.gamma", (co, co), 3)))
    bb, gb, pd = g.beta_reparam.bound_value, g.gamma_reparam.bound_value, g.beta_reparam.pedestal_value
    stem = ci < 4
    x_dtype = None if stem else BF

    def two(xin):
        if stem:
            c = FB.image_conv2d_bf16(xin, w, b, s, p)
        elif tr:
            c = FB.conv_transpose2d_bf16(xin, w, b, s, p, op)
        else:
            c = FB.conv2d_bf16(xin, w, b, s, p)
        return FB.gdn_bf16(c, g.beta, g.gamma, inverse, bb, gb, pd)

    def one(xin):
        return FB.conv_gdn_bf16(xin, w, b, g.beta, g.gamma, s, p, inverse, bb, gb, pd, transposed=tr, output_padding=op)

    F_.FORCE_IGEMM = (bm, 0, 0)
    try:
        names = set()
        F_.KERNEL_TRACE = names
        res = []
        for fn in (two, one):
            for t in (w, b, g.beta, g.gamma):
                t.grad = None
            tx = dev(x, d, x_dtype, grad=not stem)
            y = fn(tx)
            dy = dev(rb(np.random.RandomState(12).randn(*y.shape).astype(np.float32)), d, BF)
            y.backward(dy)
            res.append([host(y)] + ([] if stem else [host(tx.grad)]) + [host(t.grad) for t in (w, b, g.beta, g.gamma)])
        with torch.no_grad():   # inference: nothing but y is written
            y_inf = host(one(dev(x, d, x_dtype)))
    finally:
        F_.FORCE_IGEMM = None
        F_.KERNEL_TRACE = None
    tn = co // 64
    if stem and (k, s, p) == (5, 2, 2):
        assert f"stem_gdn_bf16_kernel<{co // 32}, {8 if co == 192 else 4}>" in names, names
    elif bm == 512 and tr:
        assert "halo_convt_bf16_kernel<2, true>" in names and "halo_convt_bf16_kernel<2, false>" in names, names
    elif bm == 512:
        assert "halo_conv_bf16_kernel<2, true, 0>" in names and "halo_conv_bf16_kernel<2, false, 0>" in names, names
    else:
        assert any(n.startswith(f"igemm_bf16_kernel<{256 if bm == 256 else 128}, {tn}, false, true") for n in names), names
    whats = ["y"] + ([] if stem else ["dx"]) + ["dw", "db", "dbeta", "dgamma"]
    # y: at most one bf16 ulp apart (an ulp is 2^-8 .. 2^-7 of the value); the direct stem also sums its 75 products
    # in another order than the column GEMM, so its conv output can itself sit one ulp away
    ya, yb = np.asarray(res[1][0], np.float64), np.asarray(res[0][0], np.float64)
    assert (np.abs(ya - yb) <= 2.0 ** -7 * np.abs(yb) + 2.0 ** -8 * 1e-2 * np.abs(yb).max()).all(), np.abs(ya - yb).max()
    assert (ya != yb).mean() < 0.05      # ... and only here and there
    for a, bq, what in zip(res[0][1:], res[1][1:], whats[1:]):
        scale_close(bq, a, 1e-2, f"{what} (fused vs conv -> gdn)")
    assert np.array_equal(y_inf, res[1][0])   # inference (y only) == training forward

@pytest.mark.parametrize("M,K,B", [(64, 3, 2), (128, 3, 2)])
def test_model_bf16_vs_fp32(env, M, K, B):
    """Config 3's model (JAH, K=3) in bf16 mode against the same weights in fp32 mode."""
    nic, FB, O, d = env
    model = nic.JointAutoregressiveHierarchical(M, K)
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    st = R.make_state(ks, 21)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(d)
    x = dev(R.make_image(B, 128, 128, 22), d)
    uz = dev(R.make_noise((B, M, 2, 2), 23), d)
    uy = dev(R.make_noise((B, M, 8, 8), 24), d)
    res, grads = {}, {}
    for prec in ("fp32", "bf16"):
        model.set_precision(prec)
        model.zero_grad(set_to_none=True)
        out = model(x, noise=(uz, uy))
        assert out["x_hat"].dtype == torch.float32 and out["y"].dtype == torch.float32
        r_ = nic.rd_loss(out, x, 0.01)
        r_["loss"].backward()
        res[prec] = r_
        grads[prec] = {k: p.grad.detach().double().flatten() for k, p in model.named_parameters()}
    for k in ("bpp_y", "bpp_z", "bpp_total", "mse"):
        a, b = res["bf16"][k], res["fp32"][k]
        assert abs(a - b) <= 0.03 * abs(b), (k, a, b)
    assert abs(res["bf16"]["psnr"] - res["fp32"]["psnr"]) <= 0.15, (res["bf16"]["psnr"], res["fp32"]["psnr"])
    worst = ("", 1.0)
    for k, g in grads["fp32"].items():
        if g.numel() < 1024:
            continue
        cs = float(torch.dot(g, grads["bf16"][k]) / (g.norm() * grads["bf16"][k].norm() + 1e-300))
        if cs < worst[1]:
            worst = (k, cs)
    assert worst[1] >= 0.98, worst
    model.set_precision("fp32")


def test_set_precision_validation(env):
    nic, FB, O, d = env
    with pytest.raises(ValueError):
        nic.JointAutoregressiveHierarchical(8, 1).set_precision("fp16")
    # (the 3x3 residual model has a bf16 mode since round 3: tests/test_gpu_variants.py)
    assert nic.HierarchicalMixtureResidual(8, 1).set_precision("bf16").encoder.precision == "bf16"
