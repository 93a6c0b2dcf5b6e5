"""Worker of tests/test_00_dp_two_rank_gpu.py (launched by torch.distributed.run, one process per rank, all
ranks on cuda:0 with the gloo back-end): data-parallel gradients of the REAL model through
parallel.GradientAllReducer -- bucket-resident gradient views written by lic_wgrad, all-reduce(mean) --
must equal the gradients of the whole batch in one process."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_image_compression_amd as nic  # noqa: E402
from neural_image_compression_amd.parallel import GradientAllReducer, broadcast_parameters, shard_batch  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    M, K, B, H = 64, 3, 4 * world, 128
    torch.manual_seed(0)
    ref = nic.JointAutoregressiveHierarchical(M, K).to(dev)
    torch.manual_seed(0 if rank == 0 else 5)      # ranks start different on purpose; broadcast fixes it
    model = nic.JointAutoregressiveHierarchical(M, K).to(dev)
    ref.overlap_branches = model.overlap_branches = False   # ranks time-share one GPU here (see bench.py)
    broadcast_parameters(model)
    for a, b in zip(ref.parameters(), model.parameters()):
        assert torch.equal(a, b), "broadcast_parameters did not replicate rank 0"
    red = GradientAllReducer(model.parameters(), bucket_mb=1.0, stream_groups=[list(model.decoder.parameters())])
    assert len(red.buckets) >= 3
    g = torch.Generator(device="cpu").manual_seed(77)
    lo, hi = shard_batch(B, rank, world)
    worst = 0.0
    for step in range(2):                           # two steps: the buckets are reused, .grad is reset to None
        x = torch.rand(B, 3, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        uz = torch.rand(B, M, H // 64, H // 64, generator=g).to(dev)
        uy = torch.rand(B, M, H // 16, H // 16, generator=g).to(dev)
        ref.zero_grad(set_to_none=True)
        nic.rd_loss(ref(x, noise=(uz, uy)), x, 0.01, sync=False)["loss"].backward()
        model.zero_grad(set_to_none=True)
        xs = x[lo:hi].contiguous(memory_format=torch.channels_last)
        out = model(xs, noise=(uz[lo:hi].contiguous(), uy[lo:hi].contiguous()))
        nic.rd_loss(out, xs, 0.01, sync=False)["loss"].backward()
        red.finish()
        torch.cuda.synchronize()
        in_bucket = 0
        for (name, p), q in zip(model.named_parameters(), ref.parameters()):
            scale = float(q.grad.abs().max())
            err = float((p.grad - q.grad).abs().max())
            worst = max(worst, err / max(scale, 1e-30))
            assert err <= 3e-4 * scale + 1e-7, (step, name, err, scale)
            lo_b = [f for f in red._flat if f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + f.numel() * 4]
            in_bucket += 1 if lo_b else 0
        assert in_bucket == len(list(model.parameters())), "every gradient must live in an all-reduce bucket"
        # the conv weight gradients were written there by the kernel itself: same storage before and after
        ids = [id(q) for q in red.params]
        for m in (model.encoder.net[0], model.encoder.net[2], model.decoder.net[0], model.decoder.net[6],
                  model.context_model.masked, model.entropy_parameters.net[0]):
            assert ids.index(id(m.weight)) not in red.copied_last_step, "a weight gradient was copied into its bucket"
    # a weight used by TWO nodes of one graph: the bucket slot is handed to the first weight-gradient
    # launch only, the second gets its own tensor and autograd adds them -- the sum, not twice the last one
    red.remove()
    from neural_image_compression_amd import layers as LY
    torch.manual_seed(3)
    conv = LY.Conv2d(16, 16, 3, stride=1, padding=1).to(dev)
    broadcast_parameters(conv)
    xs2 = torch.rand(2 * world, 16, 12, 12, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    y2 = conv(conv(xs2))
    (y2 * y2).mean().backward()                      # whole batch, no reducer
    want = [q.grad.clone() for q in conv.parameters()]
    conv.zero_grad(set_to_none=True)
    red2 = GradientAllReducer(conv.parameters(), bucket_mb=1.0)
    y2 = conv(conv(xs2[2 * rank:2 * rank + 2].contiguous(memory_format=torch.channels_last)))
    (y2 * y2).mean().backward()
    red2.finish()
    for q, w_ in zip(conv.parameters(), want):
        assert float((q.grad - w_).abs().max()) <= 3e-4 * float(w_.abs().max()) + 1e-7, "gradient of a weight used twice"
    red2.remove()
    # every rank must hold the same averaged gradients
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    got = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(got, flat)
    assert all(torch.equal(got[0], t) for t in got[1:])
    dist.barrier()
    if rank == 0:
        print(f"DP_OK world={world} worst_rel_err={worst:.2e}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
