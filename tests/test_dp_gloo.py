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


# ---- Trainer under data parallelism: validation numbers are averaged over the ranks -----------
class _ToyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(4)
        self.lin = torch.nn.Linear(6, 6)

    def forward(self, x, training=True):
        return {"x_hat": self.lin(x)}


def _toy_rd_loss(out, x, lam):
    mse = ((out["x_hat"] - x) ** 2).mean()
    return {"loss": mse, "bpp_total": float(mse.detach()) * 2.0, "psnr": -float(mse.detach())}


class _NullWriter:
    def add_scalar(self, *a):
        pass

    def close(self):
        pass


def _plateau_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neural_image_compression_amd.trainer import Trainer
        model = _ToyModel()
        opt = torch.optim.Adam(model.parameters(), lr=0.05)
        torch.manual_seed(20)
        train = [torch.randn(4, 6) for _ in range(4)]
        # disjoint validation shards on which the shared parameters behave differently: rank 0's images are
        # the training distribution (loss falls), rank 1's are scaled x8 (loss rises as the net fits rank 0's)
        val = [torch.randn(4, 6) * (1.0 if rank == 0 else 8.0) + (0.0 if rank == 0 else 3.0)]
        tr = Trainer(model, opt, [t[2 * rank:2 * rank + 2] for t in train], val, rd_loss=_toy_rd_loss, scheduler="plateau",
                     max_steps=8, val_interval=1, log_interval=1000, checkpoint_path=None, device="cpu",
                     writer=_NullWriter())
        tr.scheduler.patience = 0          # the reference's 100 validations of patience, shortened for the test
        tr.scheduler.threshold = 0.5       # "no improvement" = less than 50 % better: cuts happen within 8 steps
        tr.log_statistics = False
        losses = []
        orig = tr._validate

        def spy():
            v = orig()
            losses.append(v)
            return v
        tr._validate = spy
        tr.train()
        q.put((rank, opt.param_groups[0]["lr"], [p.detach().numpy().tolist() for p in model.parameters()], losses))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_plateau_scheduler_sees_the_same_validation_loss_on_every_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plateau_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, lr0, w0, v0), (_, lr1, w1, v1) = got
    assert v0 == v1 and len(v0) == 8          # the all-reduced validation loss is identical on both ranks
    assert lr0 == lr1 and lr0 < 0.05           # ... so the LR cuts happen at the same steps (and did happen)
    assert w0 == w1                            # ... and the replicated parameters stay bitwise identical
