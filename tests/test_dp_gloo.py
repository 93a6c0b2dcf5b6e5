"""N > 1 path on CPU: two gloo ranks, bucketed gradient all-reduce (mean) must reproduce the
big-batch gradient of one process (the only correctness pin for data parallelism).
Uses a small torch model: the reducer is model-agnostic and the HIP kernels need a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neural_image_compression_amd.parallel import GradientAllReducer, broadcast_parameters, shard_batch


def _net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                               torch.nn.Linear(16, 1))


def _loss(net, x):
    return (net(x) ** 2).mean()      # a batch mean, like every rd_loss term


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net()
        if rank == 1:                       # ranks start different on purpose; broadcast fixes it
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        broadcast_parameters(net)
        red = GradientAllReducer(net.parameters(), bucket_mb=0.0005, overlap=overlap)
        assert len(red.buckets) >= 3
        torch.manual_seed(11)
        x = torch.randn(8, 12)
        lo, hi = shard_batch(8, rank, world)
        for _ in range(2):                  # two steps: state must reset between them
            net.zero_grad(set_to_none=True)
            _loss(net, x[lo:hi]).backward()
            red.finish()
        q.put((rank, [p.grad.detach().numpy().tolist() for p in net.parameters()]))  # plain data: no fd passing
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_gradients_equal_big_batch(overlap):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = _net()
    torch.manual_seed(11)
    x = torch.randn(8, 12)
    _loss(net, x).backward()
    for g0, g1, p in zip(got[0], got[1], net.parameters()):
        g0, g1 = torch.tensor(g0), torch.tensor(g1)
        assert torch.equal(g0, g1)                                   # every rank holds the same average
        assert torch.allclose(g0, p.grad, rtol=1e-5, atol=1e-7)      # = the 1-process big-batch gradient


def test_single_process_is_a_noop():
    net = _net()
    red = GradientAllReducer(net.parameters())
    _loss(net, torch.randn(4, 12)).backward()
    g = [p.grad.clone() for p in net.parameters()]
    red.finish()
    assert all(torch.equal(a, p.grad) for a, p in zip(g, net.parameters()))
