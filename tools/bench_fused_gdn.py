#!/usr/bin/env python3
"""Forward time of conv -> GDN pairs of config 2, fused launch vs two launches (developer tool, GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import layers as LY  # noqa: E402

dev = torch.device("cuda:0")
B, M = 32, 192
cases = [("stem 256->128", LY.Conv2d(3, M, 5, stride=2, padding=2), False, (B, 3, 256, 256)),
         ("enc conv2 128->64", LY.Conv2d(M, M, 5, stride=2, padding=2), False, (B, M, 128, 128)),
         ("enc conv3 64->32", LY.Conv2d(M, M, 5, stride=2, padding=2), False, (B, M, 64, 64)),
         ("dec convT1 16->32", LY.ConvTranspose2d(M, M, 5, stride=2, padding=2, output_padding=1), True, (B, M, 16, 16)),
         ("dec convT2 32->64", LY.ConvTranspose2d(M, M, 5, stride=2, padding=2, output_padding=1), True, (B, M, 32, 32)),
         ("dec convT3 64->128", LY.ConvTranspose2d(M, M, 5, stride=2, padding=2, output_padding=1), True, (B, M, 64, 64))]
for name, conv, inv, shp in cases:
    seq = torch.nn.Sequential(conv, LY.GDN(M, inverse=inv)).to(dev)
    x = torch.randn(*shp, device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for fuse in (True, False):
        LY.FUSE_CONV_GDN = fuse
        with torch.no_grad():
            for _ in range(2):
                LY.run_fused(seq, x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                LY.run_fused(seq, x)
            e1.record()
            torch.cuda.synchronize()
        res[fuse] = e0.elapsed_time(e1) / 5
    LY.FUSE_CONV_GDN = "auto"
    print(f"{name:22s} fused {res[True]:7.3f} ms   two launches {res[False]:7.3f} ms")
