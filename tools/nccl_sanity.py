#!/usr/bin/env python3
"""Developer aid: does an RCCL ("nccl") process group come up on this box and move a gradient-sized
buffer?  Single rank (one-GPU boxes); the data-parallel step itself is covered by tests/test_dp_gloo.py."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import neural_image_compression_amd as nic  # noqa: E402
from neural_image_compression_amd.parallel import GradientAllReducer, broadcast_parameters  # noqa: E402

model = nic.JointAutoregressiveHierarchical(32, 1).to(dev)
broadcast_parameters(model)
t = torch.ones(14_000_000, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
w = dist.all_reduce(t, async_op=True)
w.wait()
torch.cuda.synchronize()
print(f"rccl ok: all_reduce of 56 MB on 1 rank {1e3 * (time.perf_counter() - t0):.2f} ms, sum {float(t.sum()):.0f}")
red = GradientAllReducer(model.parameters())
assert red.world == 1
dist.barrier()
dist.destroy_process_group()
