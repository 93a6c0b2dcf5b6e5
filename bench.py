#!/usr/bin/env python3
"""Headline benchmark: images/s of one training step (forward + rd_loss + backward + Adam step)
of the hot path on synthetic 256x256 RGB batches, batch 32 per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|3k|2h|4|5|hmr] [--no-cpu-baseline]

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...: one rank per GPU, images sharded (32 per rank, weak scaling), one RCCL all-reduce of
the gradients per step.  Rank 0 prints ONE JSON line.

Workload (config 2, BASELINE.json configs[1]): JointAutoregressiveHierarchical(192, K=1)
(the nearest reference surface to "scale hyperprior capacity 192"), B=32,
256x256, fp32, lambda=0.01, default-initialised weights (seed 0), x = rand (seed 1234+rank).
Other configs are parity-test cases that can be timed the same way: 3 = BASELINE configs[2]
(JAH(128, K=3), bf16 storage in the conv/GDN stacks), 3k / 2h = the same models in the other
precision, 4 = configs[3]'s per-GPU work, 5 = configs[4] (16x512x512).

Extra fields of the JSON line: `roofline` (dominant kernel by time, HIP events around every MFMA
launch of >= 2 GFLOP in the timed region), `kernels` (per-kernel ms and TFLOP/s of those), `analysis_hyperprior_fwd`
(the scope the north star quotes its target on, timed separately after the K steps) and
`cpu_baseline` (oracle/torch_ref.py on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto a few hardware queues; once an RCCL communicator exists (its own streams), the two
# compute streams of the model's decoder / latent-branch overlap ended up serialised unless the queue count
# is set explicitly (measured: 1286 -> 1350 img/s with RCCL initialised; any value from 2 to 16 does it).
# Must be in the environment before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = vector fp32 rate
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    # name: (model kind, M, K, B per GPU, H, W, lambda)
    "2": ("jah", 192, 1, 32, 256, 256, 0.01),
    "3": ("jah", 128, 3, 32, 256, 256, 0.01),    # BASELINE configs[2]: bf16 storage (model.set_precision)
    "3k": ("jah", 128, 3, 32, 256, 256, 0.01),   # config 3's model in fp32
    "2h": ("jah", 192, 1, 32, 256, 256, 0.01),   # config 2's model in bf16 storage (not the headline)
    "4": ("jah", 192, 3, 32, 256, 256, 0.01),
    "5": ("jah", 192, 3, 16, 512, 512, 0.01),
    "hmr": ("hmr", 192, 3, 32, 256, 256, 0.01),  # the 3x3 residual model, for the record
    "hmrh": ("hmr", 192, 3, 32, 256, 256, 0.01),  # ... in bf16 storage
}
BF16_CONFIGS = ("3", "2h", "hmrh")
BF16_MFMA_PEAK_TF = 2500.0


def cpu_baseline(M, K, H, W, lam, budget_s=25.0):
    """The CPU restatement ('port') timed on this host's cores on a bounded sample of the same
    workload: fwd + rd_loss + bwd of up to 16 images in batches of 4 through oracle/torch_ref.py
    (torch CPU ops on <=16 threads: the library the reference's own CPU path runs on) -> `value`;
    one image through the plain-C oracle (oracle/lic_oracle.c) -> `c_oracle_images_per_s`."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_recipe as R
    import neural_image_compression_amd as nic
    from oracle import oracle as O
    # a one-GPU box's CPU share is 16 cores: use at most that many OpenMP threads
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("LIC_CPU_BASELINE_THREADS", "16")))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    from oracle import torch_ref as TR
    torch.set_num_threads(cores)
    m = nic.JointAutoregressiveHierarchical(M, K)
    ks = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    st = R.make_state(ks, 0)
    # (a) the torch-op restatement on the host cores (oneDNN/MKL: the library the reference's own CPU
    #     path runs on): the fairer CPU number, reported as `value`
    B = 4 if H * W <= 256 * 256 else 1
    x = R.make_image(B, H, W, 1234)
    uz, uy = R.make_noise((B, M, H // 64, W // 64), 1), R.make_noise((B, M, H // 16, W // 16), 2)
    TR.step(st, x, M, K, "5x5", (uz, uy), lam)  # warm-up (thread pool, primitive caches)
    n, t0 = 0, time.perf_counter()
    while True:
        TR.step(st, x, M, K, "5x5", (uz, uy), lam)
        n += B
        el = time.perf_counter() - t0
        if el > budget_s * 0.6 or n >= 16:
            break
    torch_ips = n / el
    # (b) the plain-C oracle (OpenMP loops, no BLAS), one image, for the record
    O.lib()
    t0 = time.perf_counter()
    O.model_forward(dict(st), x[:1], M, K, "5x5", training=True, noise=(uz[:1], uy[:1]), lambda_rd=lam,
                    backward=True)
    c_ips = 1.0 / (time.perf_counter() - t0)
    return {"value": round(torch_ips, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} images {H}x{W} in batches of {B}, fwd+rd_loss+bwd, JAH M={M} K={K}, fp32, "
                      f"oracle/torch_ref.py (torch CPU ops, {cores} threads)",
            "c_oracle_images_per_s": round(c_ips, 4)}


def analysis_hyperprior_fwd(model, x, F_, bf16, reps=5, planned=False):
    """Forward of everything but the synthesis transform (the north star's target scope), timed
    apart from the K steps: wall time of `reps` passes, algorithmic FLOP / bytes from the launch
    plans of its MFMA kernels, and the roofline time max(t_flop, t_hbm) summed per launch."""
    with torch.no_grad():
        for _ in range(2):
            model.analysis_hyperprior(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model.analysis_hyperprior(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        F_.PROFILE = []
        model.analysis_hyperprior(x)
        torch.cuda.synchronize()
        prof, F_.PROFILE = F_.PROFILE, None
    planned_ms = None
    if planned:
        # the same forward captured once and replayed by the library (plan.ForwardPlan): what the GPU needs when Python is
        # not between the launches (the eager figure above is host-paced in the bf16 configurations: ~23 launches x ~20 us)
        from neural_image_compression_amd.plan import ForwardPlan
        fp = ForwardPlan(model, x)
        for _ in range(3):
            fp(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4 * reps):
            fp(x)
        torch.cuda.synchronize()
        planned_ms = (time.perf_counter() - t0) / (4 * reps) * 1e3
        planned_info = fp.info
        fp.close()
    flops = sum(p[1] for p in prof)
    abytes = sum(p[2] for p in prof)
    t_roof = sum(max(p[1] / ((BF16_MFMA_PEAK_TF if "bf16" in p[0] else FP32_MFMA_PEAK_TF) * 1e12),
                     p[2] / (HBM_PEAK_GBS * 1e9)) for p in prof) * 1e3
    mfma_ms = sum(p[3].elapsed_time(p[4]) for p in prof)
    return {"ms": round(ms, 3), "mfma_kernel_ms": round(mfma_ms, 3), "alg_gflop": round(flops / 1e9, 1),
            "alg_gb": round(abytes / 1e9, 3), "tflops": round(flops / ms / 1e9, 2),
            "hbm_frac": round(abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "mfma_frac": round(flops / (ms * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TF if bf16 else FP32_MFMA_PEAK_TF), 4),
            "t_roof_ms": round(t_roof, 3), "roofline_fraction": round(t_roof / ms, 4),
            "images_per_s": round(x.shape[0] / ms * 1e3, 1),
            **({"planned_ms": round(planned_ms, 3), "planned_roofline_fraction": round(t_roof / planned_ms, 4),
                "planned_launches": planned_info["kernels"]} if planned_ms is not None else {})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS))
    ap.add_argument("--plan", default="auto", choices=("auto", "on", "off"),
                    help="replay the step from a launch plan (plan.StepPlan: the captured step issued on two streams by "
                         "liblic_hip.so, no Python between launches); auto = on for the host-paced bf16 configs on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-events", action="store_true")
    ap.add_argument("--no-analysis-fwd", action="store_true",
                    help="skip the separate analysis+hyperprior forward timing (keeps a rocprofv3 trace to the K steps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # LIC_SINGLE_DEVICE=1 + LIC_DIST_BACKEND=gloo rehearse the N>1 code path on a one-GPU box
    # (all ranks share cuda:0; RCCL itself needs one device per rank)
    if os.environ.get("LIC_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("LIC_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_red = os.environ.get("LIC_FORCE_REDUCER") == "1"  # one-rank rehearsal of the RCCL data path
    if world > 1 or force_red:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import neural_image_compression_amd as nic
    from neural_image_compression_amd import functional as F_
    from neural_image_compression_amd.parallel import GradientAllReducer, broadcast_parameters

    kind, M, K, B, H, W, lam = CONFIGS[args.config]
    torch.manual_seed(0)
    model = (nic.HierarchicalMixtureResidual if kind == "hmr" else nic.JointAutoregressiveHierarchical)(M, K).to(dev)
    # Two-stream overlap of the decoder with the latent-side branch (+4 %).  Also on under RCCL data
    # parallelism: rehearsed on one rank with the reducer's hooks and collectives forced
    # (LIC_FORCE_REDUCER=1: 1322 vs 1280 img/s without overlap) -- it needs GPU_MAX_HW_QUEUES set (see the
    # top of this file).  Off for the gloo rehearsal, where several ranks time-share ONE GPU and every
    # bucket's exchange waits out the other process's time slice.  LIC_OVERLAP=0/1 forces either.
    ov = os.environ.get("LIC_OVERLAP")
    model.overlap_branches = (world == 1 or backend == "nccl") if ov is None else (ov == "1")
    bf16 = args.config in BF16_CONFIGS
    if bf16:
        model.set_precision("bf16")
    broadcast_parameters(model)
    # torch.optim.Adam's arithmetic and state dict, one launch per step (optim.FusedAdam -> lic_adam_run);
    # LIC_TORCH_ADAM=1 runs torch's own nine multi-tensor launches instead (A/B)
    opt = (torch.optim.Adam if os.environ.get("LIC_TORCH_ADAM") == "1" else nic.FusedAdam)(model.parameters(), lr=1e-4)
    reducer = GradientAllReducer(model.parameters(), overlap=os.environ.get("LIC_REDUCER_NO_OVERLAP") != "1",
                                 force=force_red, stream_groups=[list(model.decoder.parameters())],
                                 group_streams=[model.side_stream()] if model.overlap_branches else None) \
        if (world > 1 or force_red) else None
    if os.environ.get("LIC_REDUCER_NOOP") == "1":  # diagnostic: process group up, no gradient exchange
        reducer = None
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.rand(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        res = nic.rd_loss(out, x, lam, sync=False)
        res["loss"].backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return res

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The bf16 configurations are host-paced (~3 ms of Python around 0.7 ms of launches and ~3 ms of GPU work per
    # step): their step is captured once and replayed by the library's launch plan -- the same kernels, operands and
    # two streams, bit-identical losses (tests/test_gpu_plan.py) -- with the optimizer's one launch behind it.
    # The fp32 configurations are GPU-bound; data-parallel runs need autograd's hooks for the gradient exchange.
    use_plan = args.plan == "on" or (args.plan == "auto" and bf16)
    if use_plan and (world > 1 or reducer is not None):
        if args.plan == "on":
            raise SystemExit("--plan on: a launch plan replays one GPU's step (the gradient all-reduce runs from autograd hooks)")
        use_plan = False
    eager_step = step
    plan = None
    if use_plan:
        from neural_image_compression_amd.plan import StepPlan
        plan = StepPlan(model, nic.rd_loss, lam, x, tune=os.environ.get("LIC_PLAN_TUNE") == "1")

        # the synthetic batch is resident in the plan's input buffer (where a data pipeline deposits its H2D copy: the
        # eager step reads `x` in place too; handing `x` itself over would add a 25 MB device copy per step)
        plan.x.copy_(x)

        def step():
            _, res = plan.step(plan.x)
            opt.step()
            return res

    for _ in range(args.warmup):
        step()
    fence()
    prof = None
    if not args.no_profile_events and plan is None:
        # bracket only the launches that can be the dominant kernel (>= 2 GFLOP): ~60 event pairs per step
        # instead of ~400, so the profile costs the step < 0.3 %
        F_.PROFILE_MIN_FLOP = 2e9
        prof = []
    # ROCm 7.0 slows every launch of the process down once ~800 timing events are alive (measured: cfg 2h read
    # 13.1 ms/step with 20 steps bracketed, 6.5 ms with the same steps unbracketed), so the events sample the
    # timed region instead of covering it: the first timed step is always bracketed, and from its pair count a
    # stride is chosen that keeps at most MAX_LIVE_PAIRS pairs alive, spread evenly over the K steps.
    MAX_LIVE_PAIRS = 300
    stride, per_step, bracketed_steps = 1, 0, 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof is not None:
            on = i % stride == 0 and len(prof) + per_step <= MAX_LIVE_PAIRS
            F_.PROFILE = prof if on else None
            bracketed_steps += int(on)
        res = step()
        if prof is not None and i == 0:
            per_step = len(prof)
            stride = max(1, -(-(args.steps * per_step) // MAX_LIVE_PAIRS))
    fence()
    el = time.perf_counter() - t0
    final_res = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in res.items()}
    if plan is not None and not args.no_profile_events:
        # a replay runs no Python between launches, so there is nothing to record events from: the per-kernel
        # brackets come from eager steps of the same model right AFTER the timed region (same kernels and launch
        # geometry; not part of `value`)
        F_.PROFILE_MIN_FLOP = 2e9
        prof = []
        for _ in range(min(4, args.steps)):
            F_.PROFILE = prof
            eager_step()
            bracketed_steps += 1
        torch.cuda.synchronize()
    F_.PROFILE = None
    F_.PROFILE_MIN_FLOP = 0.0
    res = final_res
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())

    # a throughput number over arithmetic that went wrong is not a measurement: a non-finite loss after the timed
    # steps (a race in a kernel once showed up ONLY here, as NaNs in the two-stream step) fails the run loudly
    final_loss = float(res["loss"].detach())
    ok = torch.tensor([1.0 if math.isfinite(final_loss) else 0.0], device=dev)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # (every rank leaves together)
    if float(ok.item()) == 0.0:
        if world > 1 or force_red:
            dist.destroy_process_group()
        raise SystemExit(f"bench.py: the loss after {args.warmup + args.steps} steps is {final_loss} -- refusing to report "
                         "a throughput for a diverged / corrupted run")
    if rank == 0:
        ms = el / args.steps * 1e3
        value = world * B * args.steps / el
        line = {
            "metric": "images/sec (256x256 RGB, batch 32) fwd+bwd" if (H, W, B) == (256, 256, 32)
            else f"images/sec ({H}x{H} RGB, batch {B}) fwd+bwd",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"cfg{args.config}: {type(model).__name__}(M={M},K={K}) "
                                   f"fwd+rd_loss(lambda={lam})+bwd+Adam, {B}x3x{H}x{W} per GPU, "
                                   + ("bf16 storage / fp32 accumulate in the conv+GDN stacks, fp32 elsewhere" if bf16 else "fp32"),
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "loss": round(float(res["loss"].detach()), 6),
                       "step": "launch plan: the captured step replayed on two HIP streams by lic_plan_replay + FusedAdam "
                               f"({plan.info['kernels']} kernel nodes, {plan.info['on_side_stream']} on the second stream, "
                               f"{plan.info['events']} cross-stream events)" if plan is not None else "eager (autograd)"},
        }
        if prof:
            # per-kernel totals from HIP events recorded around every MFMA launch of the timed region
            agg = {}
            for name, flops, abytes, e0, e1 in prof:
                a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                a[0] += 1
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += flops
                a[3] += abytes
            # (the event objects are released before anything else is timed: with ~800 of them alive, ROCm 7.0
            # slowed every later launch of this process down -- the analysis forward below read 20 ms instead of 4)
            del prof[:]
            dom = max(agg, key=lambda k: agg[k][1])
            n, secs, flops, abytes = agg[dom]
            ach = flops / secs / 1e12
            # HBM bytes per launch from the committed PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
            # runs of this same command, gfx950 x2 read correction applied; see the file's "_how")
            traffic = traffic_src = None
            try:
                import glob
                suffix = {"2": "", "3": "_cfg3"}[args.config]   # (KeyError -> no committed PMC passes for this config)
                newest = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")))[-1]
                pmc_file = json.load(open(newest))
                pmc = pmc_file["kernels"]
                for kname, v in pmc.items():
                    if dom in kname:
                        traffic = v["traffic_bytes_per_launch"]
                        # which committed passes the figure comes from, and when / at which commit they were folded
                        traffic_src = (f"{os.path.relpath(newest, ROOT)} (measured {pmc_file.get('_date', 'date not recorded')}, "
                                       f"folded at commit {pmc_file.get('_commit', 'not recorded')})")
            except Exception:
                traffic = None
            peak = BF16_MFMA_PEAK_TF if "bf16" in dom else FP32_MFMA_PEAK_TF
            line["roofline"] = {
                "kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                # (PMC counters need rocprofv3 around the process: `traffic` is the figure of the newest committed
                # --pmc passes of this same command, tools/refresh_profiles.sh; every other number here is live)
                "traffic_source": traffic_src,
                "launches": n, "avg_launch_ms": round(secs / n * 1e3, 4),
                "steps_bracketed": bracketed_steps,
                "bracketed": "eager steps right after the timed region (a plan replay has no Python between launches "
                             "to record events from)" if plan is not None else "sampled steps of the timed region",
                "alg_gflop_per_launch": round(flops / n / 1e9, 3),
                "hbm_frac_of_alg_bytes": round(abytes / secs / 1e9 / HBM_PEAK_GBS, 4),
            }
            line["kernels"] = {k: {"launches": v[0], "ms_per_step": round(v[1] / bracketed_steps * 1e3, 3),
                                   "tflops": round(v[2] / max(v[1], 1e-12) / 1e12, 2)} for k, v in agg.items()}
            mfma_s = sum(v[1] for v in agg.values())
            # (sum of bracketed launches / wall time; can exceed 1: two HIP streams overlap)
            line["profiled_kernel_time_over_step_time"] = round(mfma_s / bracketed_steps / (el / args.steps), 4)
        if world == 1 and not args.no_analysis_fwd:
            line["analysis_hyperprior_fwd"] = analysis_hyperprior_fwd(model, x, F_, bf16, planned=plan is not None)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(M, K, H, W, lam)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    if world > 1 or force_red:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
