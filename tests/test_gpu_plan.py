"""Launch plans (include/lic.h lic_plan_*, plan.StepPlan): a captured step replayed on two streams by the library must
do exactly what the eager step does -- same kernels, same operands, hence the same bits -- and the generic executor
must honour every dependency of a captured fork / join."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def test_plan_replays_a_fork_join_capture():
    """a diamond captured from two streams (main: a, c, d; side: b), replayed through lic_plan_replay on two other
    streams after the input changed: the result is what the dependencies demand"""
    _need_gpu()
    from neural_image_compression_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
    side_cap = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        a = x * 2.0
        for _ in range(20):              # long enough that a missing wait would read a stale `a`
            a = a * 1.0001 + 1.0
        side_cap.wait_stream(main)
        with torch.cuda.stream(side_cap):
            b = a + 1.0
            for _ in range(10):
                b = b * 0.5 + 3.0
        c = a * 3.0
        z = torch.zeros_like(c)          # (a fill: kernel or memset node)
        c = c + z
        main.wait_stream(side_cap)
        e = b + c

    def expect(xv):
        a = xv * 2.0
        for _ in range(20):
            a = a * 1.0001 + 1.0
        b = a + 1.0
        for _ in range(10):
            b = b * 0.5 + 3.0
        return b + a * 3.0

    plan = C.c_void_p()
    L.check(lib.lic_plan_create(C.c_void_p(g.raw_cuda_graph()), C.byref(plan)), "lic_plan_create")
    info = (C.c_int64 * 7)()
    L.check(lib.lic_plan_info(plan, info), "lic_plan_info")
    nodes, kernels, memsets, memcpys, on_side, events, tuned = list(info)
    assert nodes == kernels + memsets + memcpys and memcpys == 0 and on_side >= 3 and events >= 2, list(info)
    s_main, s_side, s_side2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    sides = (C.c_void_p * 2)(s_side.cuda_stream, s_side2.cuda_stream)
    for trial in range(3):
        x.copy_(torch.arange(1 << 20, device=dev, dtype=torch.float32) * (trial + 1.5))
        torch.cuda.synchronize()
        with torch.cuda.stream(s_main):
            L.check(lib.lic_plan_replay(plan, C.c_void_p(s_main.cuda_stream), sides, 1), "replay")
            got = e.clone()              # queued on `main` after the call: ordered after ALL of the plan
        torch.cuda.synchronize()
        assert torch.equal(got, expect(x)), trial
    # the tuned schedule (whichever of the two it keeps) computes the same thing
    r = (C.c_double * 4)()
    L.check(lib.lic_plan_tune(plan, C.c_void_p(s_main.cuda_stream), sides, 2, r), "lic_plan_tune")
    torch.cuda.synchronize()
    assert r[0] > 0 and r[1] > 0 and r[2] > 0 and r[3] in (0.0, 1.0)
    x.add_(3.0)
    torch.cuda.synchronize()
    L.check(lib.lic_plan_replay(plan, C.c_void_p(s_main.cuda_stream), sides, 1), "replay")
    torch.cuda.synchronize()
    assert torch.equal(e, expect(x))
    # one stream: the same plan, serialised
    x.mul_(0.25)
    torch.cuda.synchronize()
    L.check(lib.lic_plan_replay(plan, C.c_void_p(s_main.cuda_stream), None, 0), "replay")
    torch.cuda.synchronize()
    assert torch.equal(e, expect(x))
    lib.lic_plan_destroy(plan)
    assert lib.lic_plan_create(None, C.byref(plan)) == -1     # LIC_ERR_INVALID


@pytest.mark.parametrize("precision,M,K,tune", [("bf16", 128, 3, True), ("bf16", 128, 3, False), ("fp32", 64, 1, True)])
def test_step_plan_equals_eager_training(precision, M, K, tune):
    """five optimizer steps from the same seed, eager and planned: every loss and every parameter bit for bit"""
    _need_gpu()
    import neural_image_compression_amd as nic
    from neural_image_compression_amd.plan import StepPlan
    dev = torch.device("cuda:0")
    B, H, W, lam = 4, 128, 128, 0.01
    g = torch.Generator(device="cpu").manual_seed(11)
    xs = [torch.rand(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]

    def build():
        torch.manual_seed(3)
        m = nic.JointAutoregressiveHierarchical(M, K).to(dev)
        if precision == "bf16":
            m.set_precision("bf16")
        return m, nic.FusedAdam(m.parameters(), lr=1e-3)

    ma, oa = build()
    torch.cuda.manual_seed(5)
    losses_a = []
    for i in range(5):
        oa.zero_grad(set_to_none=True)
        res = nic.rd_loss(ma(xs[i % 2]), xs[i % 2], lam, sync=False)
        res["loss"].backward()
        oa.step()
        losses_a.append(float(res["loss"].detach()))

    mb, ob = build()
    plan = StepPlan(mb, nic.rd_loss, lam, xs[0], tune=tune)
    assert (plan.tuning is not None) == tune
    assert plan.info["kernels"] > 100 and plan.info["on_side_stream"] > 10 and plan.info["events"] >= 2, plan.info
    torch.cuda.manual_seed(5)
    losses_b = []
    for i in range(5):
        if i == 2:
            ob.zero_grad(set_to_none=True)   # (a caller that clears gradients between steps: the plan restores them)
        out, res = plan.step(xs[i % 2])
        ob.step()
        losses_b.append(float(res["loss"].detach()))
    assert losses_a == losses_b, (losses_a, losses_b)
    for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        assert torch.equal(pa.detach(), pb.detach()), n
    assert out["x_hat"].shape == (B, 3, H, W) and bool(torch.isfinite(out["x_hat"]).all())
    with pytest.raises(Exception, match="captured for batches"):
        plan.step(xs[0][:2])
    # the model is still usable eagerly, on the weights the planned steps trained
    with torch.no_grad():
        ea, eb = ma(xs[0], training=False)["x_hat"], mb(xs[0], training=False)["x_hat"]
    assert torch.equal(ea, eb)
    plan.close()


def test_trainer_with_step_plan_trains_like_the_eager_trainer(tmp_path):
    """Trainer(step_plan=True): same parameters and the same logged numbers as the eager trainer after a short run that
    ends on a batch of another shape (which takes the eager step)"""
    _need_gpu()
    import neural_image_compression_amd as nic
    from neural_image_compression_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(21)
    batches = [torch.rand(4, 3, 64, 64, generator=g) for _ in range(3)] + [torch.rand(2, 3, 64, 64, generator=g)]

    class Log:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append((tag, value, step))

        def close(self):
            pass

    def run(step_plan):
        torch.manual_seed(2)
        m = nic.JointAutoregressiveHierarchical(128, 3).to(dev)
        m.set_precision("bf16")
        log = Log()
        tr = Trainer(m, nic.FusedAdam(m.parameters(), lr=1e-3), batches, rd_loss=nic.rd_loss, lambda_val=0.01, max_steps=4,
                     checkpoint_path=None, writer=log, step_plan=step_plan, log_interval=100, img_interval=100,
                     val_interval=100)
        tr.log_statistics = False
        torch.cuda.manual_seed(9)
        tr.train()
        return m, log.rows, tr

    ma, rows_a, _ = run(False)
    mb, rows_b, trb = run(True)
    assert trb._plan is not None and trb._plan.replays == 3
    for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        assert torch.equal(pa.detach(), pb.detach()), n
    la = [(t, v, s) for t, v, s in rows_a if t.startswith("losses/")]
    lb = [(t, v, s) for t, v, s in rows_b if t.startswith("losses/")]
    assert la == lb and len(la) == 4 * 8   # (the loss itself is a tensor: the eight plain numbers per step)


@pytest.mark.parametrize("precision,M,K", [("bf16", 128, 3), ("fp32", 64, 1), ("fp32", 192, 3)])
def test_deferred_reductions_change_no_bit(precision, M, K):
    """the end-of-backward batched reduction (functional.can_defer / lic_reduce_batch) against the launch-by-launch
    reductions: every gradient of a bf16 model step bit for bit -- first gradients, gradients accumulated onto existing ones
    (no deferral there), and a convolution applied twice in one graph (flush before the second use)"""
    _need_gpu()
    import neural_image_compression_amd as nic
    from neural_image_compression_amd import functional as F_
    from neural_image_compression_amd import functional_bf16 as FB
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    m = nic.JointAutoregressiveHierarchical(M, K).to(dev)
    m.set_precision(precision)
    x = torch.rand(2, 3, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
    noise = (torch.rand(2, M, 2, 2, device=dev), torch.rand(2, M, 8, 8, device=dev))

    def grads(defer, passes):
        F_.DEFER_REDUCTIONS = defer
        F_.DEFER_FP32 = defer   # (the fp32 path defers only on request: LIC_DEFER_FP32=1)
        for p in m.parameters():
            p.grad = None
        for _ in range(passes):
            nic.rd_loss(m(x, noise=noise), x, 0.01, sync=False)["loss"].backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in m.parameters()]

    try:
        for passes in (1, 2):
            ga, gb = grads(True, passes), grads(False, passes)
            for (n, _), a, b in zip(m.named_parameters(), ga, gb):
                assert torch.equal(a, b), (passes, n, float((a - b).abs().max()))
        assert not F_._PENDING_JOBS and not F_._PENDING_KEEP
        # one weight, two uses
        w = (torch.randn(128, 128, 3, 3, device=dev) * 0.05).requires_grad_(True)
        b = torch.zeros(128, device=dev, requires_grad=True)
        xin = torch.randn(2, 128, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
        if precision == "bf16":
            xin = xin.to(torch.bfloat16)

        def twice(defer):
            F_.DEFER_REDUCTIONS = F_.DEFER_FP32 = defer
            w.grad = b.grad = None
            if precision == "bf16":
                y = FB.conv2d_bf16(FB.conv2d_bf16(xin, w, b, 1, 1), w, b, 1, 1, out_f32=True)
            else:
                y = F_.conv2d(F_.conv2d(xin, w, b, 1, 1), w, b, 1, 1)
            y.square().mean().backward()
            torch.cuda.synchronize()
            return w.grad.clone(), b.grad.clone()
        (wa, ba), (wb, bb) = twice(True), twice(False)
        assert torch.equal(wa, wb) and torch.equal(ba, bb)
    finally:
        F_.DEFER_REDUCTIONS, F_.DEFER_FP32 = True, False


def test_forward_plan_equals_the_eager_forward():
    """plan.ForwardPlan (analysis + hyperprior forward, captured once, replayed by the library) against the eager call
    from the same generator state: every output tensor bit for bit, with noise and with rounding"""
    _need_gpu()
    import neural_image_compression_amd as nic
    from neural_image_compression_amd.plan import ForwardPlan
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    m = nic.JointAutoregressiveHierarchical(128, 3).to(dev)
    m.set_precision("bf16")
    xs = [torch.rand(2, 3, 128, 128, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    for training in (True, False):
        fp = ForwardPlan(m, xs[0], training=training)
        assert fp.info["kernels"] >= 15 and fp.info["memcpys"] == 0
        for x in xs:
            torch.cuda.manual_seed(3)
            with torch.no_grad():
                ref = {k: v.clone() for k, v in m.analysis_hyperprior(x, training=training).items() if torch.is_tensor(v)}
            torch.cuda.manual_seed(3)
            got = fp(x)
            torch.cuda.synchronize()
            for k, v in ref.items():
                assert torch.equal(got[k], v), (training, k)
        fp.close()
