#!/usr/bin/env python3
"""Host time per C-ABI entry point over the training step of bench.py (both the forward thread and autograd's
backward thread): every function of liblic_hip.so is wrapped with a perf_counter bracket, so the table shows how
many calls a step makes, what they cost on the host, and how much of the step's host time is NOT inside the library
(Python, autograd, the allocator).  usage: python tools/host_calls.py [config] [steps]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import neural_image_compression_amd as nic  # noqa: E402
from neural_image_compression_amd import _lib as L  # noqa: E402


class Shim:
    def __init__(self, lib):
        self._lib = lib
        self.stats = {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        st = self.stats.setdefault(name, [0, 0.0])
        pc = time.perf_counter

        def call(*a):
            t0 = pc()
            r = fn(*a)
            st[1] += pc() - t0
            st[0] += 1
            return r
        setattr(self, name, call)
        return call


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    kind, M, K, B, H, W, lam = bench.CONFIGS[cfg]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = (nic.HierarchicalMixtureResidual if kind == "hmr" else nic.JointAutoregressiveHierarchical)(M, K).to(dev)
    if cfg in bench.BF16_CONFIGS:
        model.set_precision("bf16")
    model.overlap_branches = True
    opt = nic.FusedAdam(model.parameters(), lr=1e-4)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.rand(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    phase = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0}

    def step():
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        out = model(x)
        t1 = time.perf_counter()
        res = nic.rd_loss(out, x, lam, sync=False)
        t2 = time.perf_counter()
        res["loss"].backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        phase["fwd"] += t1 - t0
        phase["loss"] += t2 - t1
        phase["bwd"] += t3 - t2
        phase["opt"] += t4 - t3

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    shim = Shim(L.load())
    L._lib = shim
    for k in phase:
        phase[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    inside = sum(v[1] for v in shim.stats.values())
    calls = sum(v[0] for v in shim.stats.values())
    print(f"cfg{cfg}: wall {wall / steps * 1e3:.3f} ms/step, host enqueue {host / steps * 1e3:.3f} ms/step "
          f"(" + ", ".join(f"{k} {v / steps * 1e3:.3f}" for k, v in phase.items()) + f"); {calls / steps:.1f} library calls/step, "
          f"{inside / steps * 1e3:.3f} ms/step inside them (incl. the shim's own ~0.3 us per call)")
    for name, (n, t) in sorted(shim.stats.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:40s} {n / steps:6.1f} calls/step {t / max(n, 1) * 1e6:7.2f} us each {t / steps * 1e3:7.3f} ms/step")


if __name__ == "__main__":
    main()
