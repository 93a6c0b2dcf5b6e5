#!/usr/bin/env python3
"""A/B: the training step of bench.py launched eagerly vs replayed from a launch plan (plan.StepPlan), same seed:
the losses of the two runs must agree bit for bit.  usage: python tools/plan_step.py [config] [steps]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import neural_image_compression_amd as nic  # noqa: E402
from neural_image_compression_amd.plan import StepPlan  # noqa: E402


def build(cfg, dev):
    kind, M, K, B, H, W, lam = bench.CONFIGS[cfg]
    torch.manual_seed(0)
    model = (nic.HierarchicalMixtureResidual if kind == "hmr" else nic.JointAutoregressiveHierarchical)(M, K).to(dev)
    model.overlap_branches = os.environ.get("LIC_OVERLAP", "1") == "1"
    if cfg in bench.BF16_CONFIGS:
        model.set_precision("bf16")
    opt = nic.FusedAdam(model.parameters(), lr=1e-4)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.rand(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    return model, opt, x, lam, B


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_enq / n * 1e3


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda", 0)

    model, opt, x, lam, B = build(cfg, dev)
    losses_e = []

    def eager():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        res = nic.rd_loss(out, x, lam, sync=False)
        res["loss"].backward()
        opt.step()
        losses_e.append(res["loss"].detach().clone())
    torch.cuda.manual_seed(7)
    for _ in range(3):
        eager()
    ms, enq = timeit(eager, steps)
    print(f"cfg{cfg} eager: {ms:7.3f} ms/step ({B / ms * 1e3:8.1f} img/s), host enqueue {enq:6.3f} ms/step", flush=True)
    le = [float(v) for v in losses_e]
    del model, opt

    model, opt, x, lam, B = build(cfg, dev)
    t0 = time.perf_counter()
    plan = StepPlan(model, nic.rd_loss, lam, x)
    torch.cuda.synchronize()
    print(f"plan built in {time.perf_counter() - t0:.2f} s: {plan.info}", flush=True)
    losses_p = []

    def planned():
        _, res = plan.step(x)
        opt.step()
        losses_p.append(res["loss"].detach().clone())
    torch.cuda.manual_seed(7)
    for _ in range(3):
        planned()
    ms, enq = timeit(planned, steps)
    print(f"cfg{cfg} plan : {ms:7.3f} ms/step ({B / ms * 1e3:8.1f} img/s), host enqueue {enq:6.3f} ms/step", flush=True)
    lp = [float(v) for v in losses_p]
    ms1, enq1 = timeit(lambda: plan.step(x), steps)
    print(f"cfg{cfg} plan, replay only (no optimizer): {ms1:7.3f} ms/step, host enqueue {enq1:6.3f} ms/step", flush=True)
    plan.streams = 1
    ms1, enq1 = timeit(lambda: plan.step(x), steps)
    print(f"cfg{cfg} plan on ONE stream, replay only : {ms1:7.3f} ms/step, host enqueue {enq1:6.3f} ms/step", flush=True)
    same = sum(a == b for a, b in zip(le, lp))
    print(f"losses equal bit for bit on {same} of {len(le)} steps; first {le[0]:.6f} / {lp[0]:.6f}, last {le[-1]:.6f} / {lp[-1]:.6f}")


if __name__ == "__main__":
    main()
