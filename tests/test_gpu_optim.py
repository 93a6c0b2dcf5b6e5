"""FusedAdam (one lic_adam_run launch per step) against torch.optim.Adam on the real model: same parameters
after several steps, interchangeable state dicts, fallback for configurations the kernel does not cover."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(wd):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    from neural_image_compression_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ma = nic.JointAutoregressiveHierarchical(64, 1).to(dev)
    mb = nic.JointAutoregressiveHierarchical(64, 1).to(dev)
    mb.load_state_dict(ma.state_dict())
    oa = torch.optim.Adam(ma.parameters(), lr=3e-3, weight_decay=wd)
    ob = FusedAdam(mb.parameters(), lr=3e-3, weight_decay=wd)
    x = torch.rand(2, 3, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
    for step in range(4):
        noise = (torch.rand(2, 64, 1, 1, device=dev), torch.rand(2, 64, 4, 4, device=dev))
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad(set_to_none=True)
            nic.rd_loss(m(x, noise=noise), x, 0.01, sync=False)["loss"].backward()
        for pa, pb in zip(ma.parameters(), mb.parameters()):   # identical inputs: identical gradients
            pb.grad.copy_(pa.grad)
        oa.step()
        ob.step()
    assert ob.fused_steps == 4
    for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        # four updates of ~lr each; the two implementations round m / denom differently (operation order / fma)
        assert float((pa.detach() - pb.detach()).abs().max()) <= 1e-5 * (4 * 3e-3) + 1e-6 * float(pa.detach().abs().max()), n
        assert pb._version >= 4    # raw-pointer updates are reported to autograd (prep.StepPrep keys on versions)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 4.0
        a, b = sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"]
        assert float((a - b).abs().max()) <= 1e-6 * max(float(a.abs().max()), 1e-12)
    # a torch.optim.Adam state dict loads into FusedAdam and vice versa
    ob.load_state_dict(sa)
    oa.load_state_dict(sb)
    # the model trained with FusedAdam sees its NEW weights in the next forward (the step preparation re-packs)
    with torch.no_grad():
        out_b = mb(x, training=False)["x_hat"]
        mb.use_step_prep = False
        assert torch.equal(out_b, mb(x, training=False)["x_hat"])
        mb.use_step_prep = True
    # amsgrad is not covered by the kernel: torch's implementation runs instead
    oc = FusedAdam(mb.parameters(), lr=1e-3, amsgrad=True)
    oc.step()
    assert oc.fused_steps == 0


def test_fused_adam_fast_path_follows_changes():
    """the cached plan of FusedAdam.step() (host fast path) must notice what it depends on: a learning-rate change
    (schedulers write group['lr']), a parameter without a gradient (torch skips it: the fused launch cannot), a loaded
    state dict -- each compared with torch.optim.Adam doing the same"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from neural_image_compression_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    pa = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (1000, 37, 4096)]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = torch.optim.Adam(pa, lr=1e-2), FusedAdam(pb, lr=1e-2)

    def both(skip=None):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = torch.randn_like(a)
            a.grad, b.grad = (None, None) if i == skip else (g, g.clone())
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * max(1.0, float(a.detach().abs().max()))

    both()
    both()                                   # planned, then the fast path
    assert ob.fused_steps == 2 and ob._fast is not None
    for o in (oa, ob):
        o.param_groups[0]["lr"] = 3e-3       # what a scheduler does
    both()
    assert ob.fused_steps == 3
    both(skip=1)                             # torch leaves parameter 1 alone; FusedAdam must not update it either
    assert ob.fused_steps == 3               # (that call went through torch's implementation)
    both()
    assert ob.fused_steps == 3               # (step counts differ between parameters now: torch's path, by design)
    import copy
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))   # new state tensors (a deep copy: load_state_dict keeps
    #                                                      references): the cached plan is dropped
    assert ob._fast is None
    both()
    both()
