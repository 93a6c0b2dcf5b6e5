#!/usr/bin/env python3
"""Analysis + hyperprior forward only (the north star's target scope), N times, for a rocprofv3 --kernel-trace
--stats run: python tools/trace_fwd.py [config] [reps].  Prints wall ms per forward."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import neural_image_compression_amd as nic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kind, M, K, B, H, W, lam = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = nic.JointAutoregressiveHierarchical(M, K).to(dev)
if cfg in bench.BF16_CONFIGS:
    model.set_precision("bf16")
x = torch.rand(B, 3, H, W, device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        model.analysis_hyperprior(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.analysis_hyperprior(x)
    torch.cuda.synchronize()
print(f"cfg{cfg}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per analysis+hyperprior forward")
