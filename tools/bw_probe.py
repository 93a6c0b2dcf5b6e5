#!/usr/bin/env python3
"""Achievable HBM bandwidth of plain streaming kernels on this GPU for the read : write mixes of the GDN
kernels (developer tool): 403 MB tensors = one [32,128,128,192] fp32 activation."""
import torch

n = 32 * 128 * 128 * 192
dev = torch.device("cuda:0")
a, b, c = (torch.randn(n, device=dev) for _ in range(3))
o1, o2 = torch.empty_like(a), torch.empty_like(a)


def t(fn, byts, name, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} {ms * 1e3:7.1f} us  {byts / ms / 1e9:6.2f} TB/s", flush=True)


B = 4.0 * n
t(lambda: a.sum(), B, "1 read (sum)")
t(lambda: o1.fill_(1.0), B, "1 write (fill)")
t(lambda: torch.mul(a, 2.0, out=o1), 2 * B, "1 read + 1 write")
t(lambda: torch.add(a, b, out=o1), 3 * B, "2 reads + 1 write")
t(lambda: (torch.mul(a, 2.0, out=o1), torch.mul(a, 3.0, out=o2)), 4 * B, "2x (1 read + 1 write)")
t(lambda: torch.addcmul(a, b, c, out=o1), 4 * B, "3 reads + 1 write")
