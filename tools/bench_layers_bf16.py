#!/usr/bin/env python3
"""Per-layer kernel timing of the bf16-storage path at config-2h / config-3 shapes (developer tool, GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import functional as F_  # noqa: E402
from neural_image_compression_amd import functional_bf16 as FB  # noqa: E402
from neural_image_compression_amd import layers as LY  # noqa: E402

dev = torch.device("cuda:0")
B = 32
M = int(os.environ.get("M", "192"))


def run(name, fn, x):
    outs = []
    for it in range(4):
        F_.PROFILE = [] if it > 0 else None
        x.grad = None
        y = fn(x)
        y.backward(torch.randn_like(y))
        torch.cuda.synchronize()
        if it > 0:
            outs.append(F_.PROFILE)
    F_.PROFILE = None
    for i in range(len(outs[0])):
        nm, flops, ab = outs[0][i][0], outs[0][i][1], outs[0][i][2]
        ms = sum(o[i][3].elapsed_time(o[i][4]) for o in outs) / len(outs)
        print(f"{name:22s} #{i} {nm:28s} {ms:8.3f} ms {flops / ms / 1e9:8.1f} TF/s  alg {ab / ms / 1e6:7.1f} GB/s")


def act(c, h):
    return torch.randn(B, c, h, h, device=dev).contiguous(memory_format=torch.channels_last).to(torch.bfloat16).requires_grad_(True)


conv = LY.Conv2d(M, M, 5, stride=2, padding=2).to(dev)
convT = LY.ConvTranspose2d(M, M, 5, stride=2, padding=2, output_padding=1).to(dev)
run("enc conv2 128->64", lambda t: conv(t, bf16=True), act(M, 128))
run("enc conv3 64->32", lambda t: conv(t, bf16=True), act(M, 64))
run("dec convT3 64->128", lambda t: convT(t, bf16=True), act(M, 64))
for h in (128, 64):
    g = LY.GDN(M).to(dev)
    run(f"gdn {h}", lambda t: g(t, bf16=True), act(M, h))
