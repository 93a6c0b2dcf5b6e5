#!/usr/bin/env python3
"""cProfile of the host side of bench.py's training step (where do the ~17 us per launch go?).
usage: python tools/host_profile.py [config] [steps]"""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import neural_image_compression_amd as nic  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
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

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        res = nic.rd_loss(out, x, lam, sync=False)
        res["loss"].backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if os.environ.get("LIC_PROFILE_BACKWARD", "1") == "1":
        # autograd runs CUDA backward nodes on its own thread, which cProfile does not see: keep them on this one
        torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(70)
    st.sort_stats("cumulative").print_stats(60)


if __name__ == "__main__":
    main()
