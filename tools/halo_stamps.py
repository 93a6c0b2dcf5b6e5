#!/usr/bin/env python3
"""In-kernel stamps of halo_conv_bf16_kernel (debug build with -DLIC_HALO_ABLATE only): per workgroup
prologue / main loop / epilogue in shader cycles and in 100 MHz real time -> the clock the chip holds in the loop."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import _lib as L  # noqa: E402
from neural_image_compression_amd import functional as F_  # noqa: E402
from neural_image_compression_amd import functional_bf16 as FB  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(32, M, 128, 128, generator=g).to(dev).contiguous(memory_format=torch.channels_last).to(torch.bfloat16)
w = (torch.randn(M, M, 5, 5, generator=g) / (5.0 * M ** 0.5)).to(dev)
b = torch.randn(M, generator=g).to(dev)
F_.FORCE_IGEMM = (512, 0, 1)
with torch.no_grad():
    for _ in range(30):   # warm clocks
        FB.conv2d_bf16(x, w, b, 2, 2)
    torch.cuda.synchronize()
lib = L.load()
lib.lic_halo_debug_read.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((256, 10), np.uint64)
rc = lib.lic_halo_debug_read(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
cyc = buf[:, :5].astype(np.int64)
rt = buf[:, 5:].astype(np.int64)
for name, i, j in (("prologue", 0, 1), ("main loop (tile 1)", 1, 2), ("epilogue (tile 1)", 2, 3), ("rest", 3, 4), ("whole", 0, 4)):
    dc = np.median(cyc[:, j] - cyc[:, i])
    dr = np.median(rt[:, j] - rt[:, i]) * 10.0   # ns
    print(f"{name:20s} median {dc:10.0f} cycles {dr / 1e3:8.2f} us  -> {dc / max(dr, 1):.2f} GHz")
t0 = rt[:, 0].min()
print("workgroup start (us): max %.1f; end: min %.1f median %.1f max %.1f" % (
    ((rt[:, 0] - t0) * 1e-2).max(), ((rt[:, 4] - t0) * 1e-2).min(), np.median((rt[:, 4] - t0) * 1e-2), ((rt[:, 4] - t0) * 1e-2).max()))
