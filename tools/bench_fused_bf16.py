#!/usr/bin/env python3
"""Forward-only timing of the bf16 analysis layers of config 3 (developer tool, GPU only):
python tools/bench_fused_bf16.py [M]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import functional_bf16 as FB  # noqa: E402
from neural_image_compression_amd import layers as LY  # noqa: E402

dev = torch.device("cuda:0")
B = 32
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timeit(name, fn, reps=30):
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
    print(f"{name:34s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us")


def act(c, h):
    return torch.randn(B, c, h, h, device=dev).contiguous(memory_format=torch.channels_last).to(torch.bfloat16)


img = torch.rand(B, 3, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
stem = LY.Conv2d(3, M, 5, stride=2, padding=2).to(dev)
conv = LY.Conv2d(M, M, 5, stride=2, padding=2).to(dev)
g = LY.GDN(M).to(dev)
gd = (g.beta, g.gamma, False, g.beta_reparam.bound_value, g.gamma_reparam.bound_value, g.beta_reparam.pedestal_value)
col, wpk, _ = FB._stem_columns_bf16(img, stem.weight, 2, 2)
timeit("stem im2col", lambda: FB._stem_columns_bf16(img, stem.weight, 2, 2))
timeit("stem conv (im2col + gemm)", lambda: stem(img, bf16=True))
timeit("stem conv+gdn fused (incl im2col)", lambda: FB.conv_gdn_bf16(img, stem.weight, stem.bias, g.beta, g.gamma, 2, 2, *gd[2:]))
for h in (128, 64, 32):
    x = act(M, h)
    timeit(f"gdn {h}", lambda: g(x, bf16=True))
    timeit(f"conv {h}->{h // 2}", lambda: conv(x, bf16=True))
    timeit(f"conv+gdn fused {h}->{h // 2}", lambda: FB.conv_gdn_bf16(x, conv.weight, conv.bias, g.beta, g.gamma, 2, 2, *gd[2:]))
