#!/usr/bin/env python3
"""Halo-resident 5x5 stride-2 bf16 convolution (lic_halo_bf16.h) against the implicit-GEMM tiles: equality within
fp32 summation order and forward timing on the real layer shapes (developer tool, GPU only):
python tools/bench_halo.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import functional as F_  # noqa: E402
from neural_image_compression_amd import functional_bf16 as FB  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, reps=20):
    with torch.no_grad():
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


def run(B, M, H, W, force, transposed=False):
    g = torch.Generator(device="cpu").manual_seed(H * 7 + M)
    x = torch.randn(B, M, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).to(BF)
    w = (torch.randn(M, M, 5, 5, generator=g) / (5.0 * M ** 0.5)).to(dev)
    b = torch.randn(M, generator=g).to(dev)
    outs, times = {}, {}
    fn = (lambda: FB.conv_transpose2d_bf16(x, w, b, 2, 2, 1)) if transposed else (lambda: FB.conv2d_bf16(x, w, b, 2, 2))
    for name, f in force.items():
        F_.FORCE_IGEMM = f
        try:
            with torch.no_grad():
                outs[name] = fn().float()
            times[name] = timeit(fn)
        finally:
            F_.FORCE_IGEMM = None
    ref = outs["igemm128"]
    flop = 2.0 * B * H * W * 25 * M * M if transposed else 2.0 * B * ((H + 1) // 2) * ((W + 1) // 2) * 25 * M * M
    for name in outs:
        err = (outs[name] - ref).abs().max().item()
        print(f"{'convT' if transposed else 'conv '} B={B} M={M} {H}x{W} {name:10s} {times[name]:8.1f} us {flop / times[name] * 1e-6:8.1f} TF  "
              f"max|d| vs igemm128 {err:.3e} (scale {ref.abs().max().item():.2f})", flush=True)


if __name__ == "__main__":
    Ms = [int(a) for a in sys.argv[1:]] or [128, 192]
    force = {"igemm128": (128, 0, 1), "auto": None, "halo": (512, 0, 1)}
    run(2, 128, 37, 45, force)      # ragged: partial tiles in both directions
    for M in Ms:
        run(32, M, 128, 128, force)
        run(32, M, 64, 64, force)
        run(32, M, 64, 64, force, transposed=True)
        run(32, M, 32, 32, force, transposed=True)
