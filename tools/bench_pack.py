#!/usr/bin/env python3
"""Weight packing: LDS-tiled kernel vs the generic gather kernel, per conv-weight shape (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import functional as F_  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (d0, d1, k) in ((192, 192, 5), (128, 128, 5), (384, 192, 5), (384, 288, 3), (192, 192, 3), (640, 768, 1), (192, 3, 5)):
    w = torch.randn(d0, d1, k, k, device=dev)
    row = f"[{d0},{d1},{k},{k}]"
    for dg in (False, True):
        os.environ.pop("LIC_PACK_NO_TILED", None)
        a = t(lambda: F_._pack_conv_weight(w, False, dg))
        os.environ["LIC_PACK_NO_TILED"] = "1"
        b = t(lambda: F_._pack_conv_weight(w, False, dg))
        row += f"  {'dgrad' if dg else 'fwd  '}: tiled {a:6.1f} us, generic {b:6.1f} us"
    print(row, flush=True)
