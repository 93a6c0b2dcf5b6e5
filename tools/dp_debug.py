#!/usr/bin/env python3
"""Developer aid: where does a 2-rank step on ONE shared GPU spend its time (gloo rehearsal)?"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_image_compression_amd as nic  # noqa: E402
from neural_image_compression_amd.parallel import GradientAllReducer  # noqa: E402

dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = nic.JointAutoregressiveHierarchical(192, 1).to(dev)
model.overlap_branches = os.environ.get("LIC_NO_OVERLAP") != "1"
x = torch.rand(32, 3, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
red = GradientAllReducer(model.parameters(), overlap=False)


def t(fn, n=4):
    fn()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwdbwd():
    model.zero_grad(set_to_none=True)
    out = model(x)
    nic.rd_loss(out, x, 0.01, sync=False)["loss"].backward()


def full():
    fwdbwd()
    red.finish()


def pieces():
    """red.finish() taken apart, host-timed with a device sync after each piece"""
    fwdbwd()
    torch.cuda.synchronize()
    out = {}
    t0 = time.perf_counter()
    ps = [p for p in model.parameters()]
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    torch.cuda.synchronize()
    out["cat"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    w = dist.all_reduce(flat, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    out["all_reduce(fresh flat)"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    off = 0
    for p in ps:
        n = p.numel()
        torch.mul(flat[off:off + n].view_as(p), 0.5, out=p.grad)
        off += n
    torch.cuda.synchronize()
    out["scatter"] = (time.perf_counter() - t0) * 1e3
    return out


pieces()
pc = pieces()
if rank == 0:
    print("pieces:", {k: round(v, 1) for k, v in pc.items()}, flush=True)
a = t(fwdbwd)
b = t(full)
flat = torch.zeros(14_000_000, device=dev)
c = t(lambda: dist.all_reduce(flat))
if rank == 0:
    print(f"overlap={model.overlap_branches}: fwd+bwd {a:.1f} ms, +allreduce {b:.1f} ms, bare 56 MB all_reduce {c:.1f} ms", flush=True)
dist.destroy_process_group()
