#!/usr/bin/env python3
"""lic_gdn_bwd_bf16 (one sweep) alone against the two-launch route: time and the HBM rate of its algorithmic bytes
(5 tensors of P x C bf16).  usage: python tools/bench_gdn_bwd_bf16.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from neural_image_compression_amd import _lib as L  # noqa: E402
from neural_image_compression_amd import functional_bf16 as FB  # noqa: E402
from neural_image_compression_amd.functional import _ptr, _stream  # noqa: E402

d = torch.device("cuda:0")
BF = torch.bfloat16
lib = L.load()


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for C, side in ((128, 128), (128, 64), (128, 32), (64, 128)):
    P = 32 * side * side
    x = torch.randn(P, C, device=d).to(BF)
    g = torch.randn(P, C, device=d).to(BF)
    n = (torch.rand(P, C, device=d) + 0.5).to(BF)
    ge = torch.rand(C, C, device=d) * 0.05
    gp1 = FB._pack_bf16(ge, 1, C, C, 0, C, 1, kperm=True)
    gp2 = FB._pack_bf16(ge, 1, C, C, 0, C, 1)
    dx, t = torch.empty_like(x), torch.empty_like(x)

    def one():
        L.check(lib.lic_gdn_bwd_bf16(_ptr(g), _ptr(x), _ptr(n), _ptr(gp1), _ptr(dx), _ptr(t), None, None, P, C, 0, _stream()), "x")

    def two():
        L.check(lib.lic_gdn_dnorm_bf16(_ptr(g), _ptr(x), _ptr(n), _ptr(t), x.numel(), 0, _stream()), "dnorm")
        FB._igemm_bf16(t, gp2, dx, B=1, Hi=1, Wi=P, Cin=C, Ho=1, Wo=P, Cout=C, kh=1, kw=1, stride=1, pad=0,
                       transposed=False, epilogue=L.EPI_GDN_BWD, aux=g, aux2=x, aux3=n)
    t1, t2 = timed(one), timed(two)
    gb = 5 * P * C * 2 / 1e9
    print(f"C={C} {side}x{side}x32: one sweep {t1:7.1f} us ({gb / t1 * 1e6 / 1e3:5.2f} TB/s of {gb * 1e3:.0f} MB), two launches {t2:7.1f} us")
