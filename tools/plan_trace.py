#!/usr/bin/env python3
"""Ten planned steps (plan.StepPlan + FusedAdam) and nothing else: the process to put under
`rocprofv3 --kernel-trace` when the question is what the replay looks like on the GPU (tools/timeline.py reads the
database).  usage: python tools/plan_trace.py [config] [steps] [streams]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import neural_image_compression_amd as nic  # noqa: E402
from neural_image_compression_amd.plan import StepPlan  # noqa: E402
from plan_step import build, timeit  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
model, opt, x, lam, B = build(cfg, dev)
plan = StepPlan(model, nic.rd_loss, lam, x, tune=os.environ.get("LIC_PLAN_TUNE", "0") == "1")
plan.streams = int(sys.argv[3]) if len(sys.argv) > 3 else 3


def step():
    plan.step(x)
    opt.step()


mode = sys.argv[4] if len(sys.argv) > 4 else "null"
if mode == "sync":       # no run-ahead of the host
    inner = step

    def step():
        inner()
        torch.cuda.synchronize()
ctx = torch.cuda.stream(torch.cuda.Stream()) if mode == "pool" else torch.cuda.stream(torch.cuda.current_stream())
ctx.__enter__()
for _ in range(3):
    step()
ms, enq = timeit(step, steps)
print(mode, end=": ")
print(f"cfg{cfg} plan ({plan.streams} streams): {ms:.3f} ms/step, host enqueue {enq:.3f} ms/step; {plan.info} {plan.tuning}")
