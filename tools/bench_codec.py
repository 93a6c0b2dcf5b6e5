#!/usr/bin/env python3
"""Wall time of ContextCodec.compress / decompress on a Kodak-sized image (GPU only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_image_compression_amd as nic  # noqa: E402
from neural_image_compression_amd.codec import ContextCodec  # noqa: E402

M, K = int(os.environ.get("M", "192")), int(os.environ.get("K", "3"))
H, W = int(os.environ.get("H", "512")), int(os.environ.get("W", "768"))
torch.manual_seed(0)
model = nic.JointAutoregressiveHierarchical(M, K).cuda().eval()
x = torch.rand(1, 3, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
cc = ContextCodec(model)
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = cc.compress(x)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dec = cc.decompress(enc["strings"], enc["shape"], enc["z_shape"])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ok = torch.equal(dec["y_hat"], enc["y_in"])
    npx = enc["shape"][2] * enc["shape"][3]
    print(f"JAH({M},{K}) {H}x{W}: compress {1e3 * (t1 - t0):8.1f} ms, decompress {1e3 * (t2 - t1):8.1f} ms "
          f"({1e6 * (t2 - t1) / npx:6.1f} us per latent pixel, {npx} pixels), round trip {'ok' if ok else 'MISMATCH'}, "
          f"bpp coded {enc['bpp_coded']:.4f} est {enc['bpp_est']:.4f}", flush=True)
