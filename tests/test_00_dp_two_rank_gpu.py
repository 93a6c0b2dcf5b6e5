"""Two data-parallel ranks of the REAL model on one MI355X (gloo back-end, both ranks on cuda:0): the
all-reduced gradients must equal the one-process whole-batch gradients (tests/dp_worker.py).

The ranks are separate programs started with torch.distributed.run, i.e. with an exec in a child process,
and that must happen before THIS process has initialised the GPU (a GPU box refuses the exec afterwards).
The file name sorts first and tests/conftest.py does not touch the GPU at collection, so under
`pytest tests -m gpu` this test runs before any other GPU test; started later it skips."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_real_model_gradients_equal_whole_batch():
    if torch.cuda.device_count() == 0:
        pytest.skip("needs an MI355X")
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: the rank programs must be started before that "
                    "(run `pytest tests -m gpu`, where this file comes first, or this file alone)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "DP_OK world=2" in text, text[-4000:]


def _run_bench(extra_env, launcher=None, args=("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-analysis-fwd",
                                              "--no-profile-events")):
    """bench.py as a child process (started before this process touches the GPU); returns its JSON line"""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), **extra_env)
    cmd = [sys.executable] + (launcher(port) if launcher else []) + [os.path.join(ROOT, "bench.py")] + list(args)
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = r.stdout.decode(errors="replace")
    assert r.returncode == 0, text[-4000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, text[-4000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_prints_one_line():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on a
    one-GPU box: both ranks on cuda:0, gloo instead of RCCL.  One JSON line, from rank 0, for the whole job."""
    if torch.cuda.device_count() == 0:
        pytest.skip("needs an MI355X")
    if torch.cuda.is_initialized():
        pytest.skip("the rank programs must be started before this process initialises the GPU")
    line = _run_bench({"LIC_SINGLE_DEVICE": "1", "LIC_DIST_BACKEND": "gloo"},
                      launcher=lambda port: ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                             "--master-addr", "127.0.0.1", "--master-port", str(port)],
                      args=("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-analysis-fwd",
                            "--no-profile-events"))
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["config"]["parallelism"] == "dp2"
    assert line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    import math
    assert math.isfinite(line["config"]["loss"]) and line["value"] > 0


@pytest.mark.gpu
def test_one_rank_rccl_reducer_gives_the_plain_run_s_loss():
    """The RCCL-only branches of parallel.GradientAllReducer (ReduceOp.AVG, hooks registered under the side stream,
    no per-gradient fences) exercised on the one GPU there is: `LIC_FORCE_REDUCER=1 bench.py` runs the hooks and the
    collectives on a one-rank RCCL communicator.  A mean over one rank changes nothing, so the loss after the same
    steps must be bit-identical to the plain run's."""
    if torch.cuda.device_count() == 0:
        pytest.skip("needs an MI355X")
    if torch.cuda.is_initialized():
        pytest.skip("the bench programs must be started before this process initialises the GPU")
    plain = _run_bench({})
    forced = _run_bench({"LIC_FORCE_REDUCER": "1"})
    assert forced["n_gpus"] == 1 and plain["n_gpus"] == 1
    assert forced["config"]["loss"] == plain["config"]["loss"], (forced["config"]["loss"], plain["config"]["loss"])
