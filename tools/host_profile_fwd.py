#!/usr/bin/env python3
"""cProfile of the HOST side of the analysis+synthesis forward under no_grad (all on the main thread), sorted by
own time and by cumulative time.  usage: python tools/host_profile_fwd.py [config] [reps]"""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import neural_image_compression_amd as nic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kind, M, K, B, H, W, lam = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = nic.JointAutoregressiveHierarchical(M, K).to(dev)
if cfg in bench.BF16_CONFIGS:
    model.set_precision("bf16")
x = torch.rand(B, 3, H, W, device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host enqueue {(t1 - t0) / reps * 1e3:.3f} ms per forward, with GPU {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        model(x)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(25)
