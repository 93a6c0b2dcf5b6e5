#!/usr/bin/env python3
"""A/B: the training step of bench.py launched eagerly vs replayed from one captured HIP graph.
usage: python tools/graph_step.py [config] [steps]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import neural_image_compression_amd as nic  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    kind, M, K, B, H, W, lam = bench.CONFIGS[cfg]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = (nic.HierarchicalMixtureResidual if kind == "hmr" else nic.JointAutoregressiveHierarchical)(M, K).to(dev)
    model.overlap_branches = os.environ.get("LIC_OVERLAP", "1") == "1"
    if cfg in bench.BF16_CONFIGS:
        model.set_precision("bf16")
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.rand(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        res = nic.rd_loss(out, x, lam, sync=False)
        res["loss"].backward()
        opt.step()
        return res["loss"].detach()

    def timeit(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, t_enq / n * 1e3

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    ms, enq = timeit(step, steps)
    print(f"cfg{cfg} eager : {ms:7.3f} ms/step ({B / ms * 1e3:8.1f} img/s), host enqueue {enq:6.3f} ms/step", flush=True)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    l0 = float(loss)
    ms, enq = timeit(graph.replay, steps)
    print(f"cfg{cfg} graph : {ms:7.3f} ms/step ({B / ms * 1e3:8.1f} img/s), host enqueue {enq:6.3f} ms/step; "
          f"loss after first replay {l0:.6f}, now {float(loss):.6f}", flush=True)


if __name__ == "__main__":
    main()
