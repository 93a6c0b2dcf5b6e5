#!/usr/bin/env python3
"""The RGB layers of config 3 in bf16 storage, alone: stem (+GDN) forward / backward and head forward / backward, with
the direct kernels and with the column-matrix route (LIC_BF16_HEAD_DIRECT=0), timed with HIP events around the whole
forward and the whole backward of each layer.  usage: python tools/bench_rgb_bf16.py [C] [B] [H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from neural_image_compression_amd import functional_bf16 as FB  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
d = torch.device("cuda:0")
BF = torch.bfloat16


def timed(fn, reps=20):
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


def head(direct):
    os.environ["LIC_BF16_HEAD_DIRECT"] = "1" if direct else "0"
    x = torch.randn(B, C, H // 2, H // 2, device=d).to(BF).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(C, 3, 5, 5, device=d) * 0.02).requires_grad_(True)
    b = torch.zeros(3, device=d, requires_grad=True)
    g = torch.randn(B, 3, H, H, device=d).contiguous(memory_format=torch.channels_last)
    out = [None]

    def fwd():
        out[0] = FB.image_conv_transpose2d_bf16(x, w, b, 2, 2, 1)

    def fb():
        x.grad = w.grad = b.grad = None
        FB.image_conv_transpose2d_bf16(x, w, b, 2, 2, 1).backward(g)
    tf, tfb = timed(fwd), timed(fb)
    print(f"head C={C} {'direct ' if direct else 'columns'}: forward {tf:7.1f} us, forward+backward {tfb:7.1f} us (backward {tfb - tf:7.1f} us)")


head(True)
head(False)
