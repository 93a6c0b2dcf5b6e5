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
