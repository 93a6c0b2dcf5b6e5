#!/usr/bin/env python3
"""How fast is a gloo all-reduce of a gradient-sized buffer between two ranks sharing one GPU?
(CUDA tensor through gloo's own staging vs an explicit pinned host copy.)  Launch with torch.distributed.run."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.randn(4 * 1024 * 1024, device="cuda")  # 16 MB
h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
for name in ("cuda tensor", "pinned host copy", "cuda tensor", "pinned host copy"):
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        if name == "cuda tensor":
            dist.all_reduce(x)
        else:
            h.copy_(x, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            dist.all_reduce(h)
            x.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if r == 0:
        print(f"{name:18s}: {dt / 4 * 1e3:8.1f} ms per 16 MB all-reduce", flush=True)
dist.destroy_process_group()
