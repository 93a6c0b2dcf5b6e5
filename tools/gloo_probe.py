#!/usr/bin/env python3
"""How long does a gloo all-reduce take between two ranks that SHARE one GPU, as a function of the buffer size
and of whether the GPU is busy with the ranks' own kernels?  (The data-parallel rehearsal on a one-GPU box;
explains why an extra, latency-sized tail bucket made that rehearsal 10-100x slower: DESIGN.md section 5.)
Launch:  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/gloo_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
busy_a = torch.randn(8192, 8192, device="cuda")


def timed(n_floats, busy, reps=4):
    x = torch.randn(n_floats, device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        if busy:                       # ~20 ms of queued GPU work in front of the collective, in BOTH ranks
            for _ in range(4):
                busy_a @ busy_a
        dist.all_reduce(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for busy in (False, True):
    for n in (256, 16 * 1024, 256 * 1024, 4 * 1024 * 1024):
        ms = timed(n, busy)
        if r == 0:
            print(f"gloo all-reduce of {n * 4 / 1024:9.1f} KiB, GPU {'busy' if busy else 'idle'}: {ms:8.2f} ms", flush=True)
dist.destroy_process_group()
