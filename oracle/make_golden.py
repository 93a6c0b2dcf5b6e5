#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own Python modules.

Run in the build container only (needs /root/reference):   python oracle/make_golden.py
The reference never travels to the GPU box; the fixtures (data only) do.

What is imported from /root/reference as-is: utils, EntropyModels, ParametersModels,
ContextModels, RateDistortionLoss (op-level fixtures: fully reference-pinned), and -- for the
whole-model fixtures -- Layers, Components, Models.  Those three import the third-party
`compressai.layers.gdn.GDN`, which is absent offline; an in-memory module implementing the
public CompressAI GDN definition is registered under that name first, so
whole-model fixtures pin everything EXCEPT the GDN arithmetic ("GDN-unpinned").

Parameters / inputs / noise come from tests/golden_recipe.py (numpy only) and are loaded into
the reference modules with load_state_dict, so fixtures store expected outputs only.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
import golden_recipe as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(4)
torch.manual_seed(0)


# --- stand-in for the absent third-party package (public CompressAI definition) ----------------
class _LowerBoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)).type(g.dtype) * g, None


class _LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


class _NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        pedestal = float(reparam_offset) ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = _LowerBound((float(minimum) + pedestal) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        return self.lower_bound(x) ** 2 - self.pedestal


class _GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = _NonNegativeParametrizer(minimum=beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = _NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma_init * torch.eye(in_channels)))

    def forward(self, x):
        C = x.size(1)
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma).reshape(C, C, 1, 1)
        norm = F.conv2d(x ** 2, gamma, beta)
        norm = torch.sqrt(norm) if self.inverse else torch.rsqrt(norm)
        return x * norm


def _register_gdn_standin():
    pkg = types.ModuleType("compressai")
    layers = types.ModuleType("compressai.layers")
    gdn = types.ModuleType("compressai.layers.gdn")
    gdn.GDN = _GDN
    layers.gdn = gdn
    layers.GDN = _GDN
    pkg.layers = layers
    sys.modules["compressai"] = pkg
    sys.modules["compressai.layers"] = layers
    sys.modules["compressai.layers.gdn"] = gdn


# --- helpers ----------------------------------------------------------------------------------
def T(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if grad:
        t.requires_grad_(True)
    return t


def load_recipe_state(mod: nn.Module, seed: int):
    ks = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
    st = R.make_state(ks, seed)
    mod.load_state_dict({k: T(v) for k, v in st.items()})
    return ks


def ks_json(ks):
    return np.array(json.dumps([[k, list(s)] for k, s in ks]))


def subsample(g: np.ndarray) -> np.ndarray:
    g = np.asarray(g, np.float32).ravel()
    if g.size > 8192:
        g = g[:: -(-g.size // 4096)]
    return g.copy()


def save(name, **kw):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --- op-level, fully reference-pinned -----------------------------------------------------------
def gen_entropy_parameters(K):
    from ParametersModels import EntropyParameters
    M, B, h, w = 8, 2, 4, 6
    m = EntropyParameters(M, M, K)
    ks = load_recipe_state(m, seed=11 + K)
    comb = R.make_noise((B, 4 * M, h, w), 21 + K) * 4 - 2
    x = T(comb, True)
    outs = m(x)
    cots = [R.make_noise(tuple(o.shape), 31 + i) - 0.5 for i, o in enumerate(outs)]
    loss = sum((o * T(c)).sum() for o, c in zip(outs, cots))
    loss.backward()
    kw = {"keys_shapes": ks_json(ks), "dx": x.grad.numpy()}
    for i, o in enumerate(outs):
        kw[f"out{i}"] = o.detach().numpy()
    for k, p in m.named_parameters():
        kw["grad." + k] = subsample(p.grad.numpy())
    save(f"op_entropy_parameters_K{K}.npz", M=M, K=K, B=B, h=h, w=w, seed_state=11 + K,
         seed_in=21 + K, seed_cot=31, **kw)


def gen_factorized():
    from EntropyModels import FactorizedEntropyBottleneck
    C, B, h, w = 6, 3, 4, 5
    m = FactorizedEntropyBottleneck(C)
    ks = load_recipe_state(m, seed=41)
    xin = (R.make_noise((B, C, h, w), 42) * 12 - 6).astype(np.float32)
    xin[0] = np.round(xin[0])  # integer centres (eval mode)
    xin[1, 0, 0, :3] = [3000.0, -3000.0, 0.0]  # saturating tails -> clamp at 1e-9
    x = T(xin, True)
    p = m(x)
    cot = R.make_noise(tuple(p.shape), 43) - 0.5
    (torch.log(p) * T(cot)).sum().backward()
    kw = {"keys_shapes": ks_json(ks), "x": xin, "p": p.detach().numpy(), "dx": x.grad.numpy()}
    for k, q in m.named_parameters():
        kw["grad." + k] = q.grad.numpy()
    xs = np.linspace(-8, 8, 33).astype(np.float32)
    kw["xs"] = xs
    kw["cdf_ch2"] = m.channel_cdf(2, T(xs)).numpy()
    kw["pmf_ch2"] = m.channel_pmf(2, T(xs)).numpy()
    # sign-trick edge (EntropyModels.py:138-143): p for a raw likelihood pass (no clamp)
    kw["p_raw"] = m._likelihood(T(xin)).detach().numpy()
    save("op_factorized.npz", C=C, B=B, h=h, w=w, seed_state=41, seed_cot=43, **kw)


def gen_gaussian():
    from EntropyModels import GaussianConditional, GaussianMixtureConditional
    B, M, h, w, K = 2, 5, 4, 4, 3
    r = np.random.RandomState(51)
    x = np.round(r.randn(B, M, h, w) * 3).astype(np.float32)
    x[1] += r.rand(M, h, w).astype(np.float32) - 0.5
    mu = (r.randn(B, M, h, w) * 2).astype(np.float32)
    sg = (np.abs(r.randn(B, M, h, w)) * 1.5 + 0.05).astype(np.float32)
    # tails: erf-difference cancellation and the 1e-9 clamp region
    x[0, 0, 0, :4] = [30.0, -30.0, 6.0, -6.0]
    mu[0, 0, 0, :4] = 0.0
    sg[0, 0, 0, :4] = [1.0, 1.0, 1.0, 0.3]
    sg[0, 1, 0, 0] = 1e-6  # softplus floor
    g1 = GaussianConditional()
    tx, tm, ts = T(x, True), T(mu, True), T(sg, True)
    p1 = g1(tx, mu=tm, sigma=ts)
    cot = R.make_noise(tuple(p1.shape), 52) - 0.5
    (torch.log(p1) * T(cot)).sum().backward()
    kw = dict(x=x, mu=mu, sigma=sg, p1=p1.detach().numpy(), dx1=tx.grad.numpy(),
              dmu1=tm.grad.numpy(), dsigma1=ts.grad.numpy())
    mus = (r.randn(B, K, M, h, w) * 2).astype(np.float32)
    sgs = (np.abs(r.randn(B, K, M, h, w)) * 1.5 + 0.05).astype(np.float32)
    wl = r.randn(B, K, M, h, w).astype(np.float32)
    ws = np.exp(wl) / np.exp(wl).sum(1, keepdims=True)
    ws = ws.astype(np.float32)
    g3 = GaussianMixtureConditional()
    tx, tw, tm, ts = T(x, True), T(ws, True), T(mus, True), T(sgs, True)
    p3 = g3(tx, weights=tw, mus=tm, sigmas=ts)
    (torch.log(p3) * T(cot)).sum().backward()
    kw.update(weights=ws, mus=mus, sigmas=sgs, p3=p3.detach().numpy(), dx3=tx.grad.numpy(),
              dw3=tw.grad.numpy(), dmu3=tm.grad.numpy(), dsigma3=ts.grad.numpy())
    save("op_gaussian.npz", seed_cot=52, **kw)


def gen_rd_loss():
    from RateDistortionLoss import rd_loss
    B = 3
    r = np.random.RandomState(61)
    logp_y = np.log(r.uniform(1e-9, 1.0, (B, 8, 4, 6))).astype(np.float32)
    logp_z = np.log(r.uniform(1e-3, 1.0, (B, 8, 1, 2))).astype(np.float32)
    x = r.rand(B, 3, 16, 24).astype(np.float32)
    xh = (x + 0.05 * r.randn(B, 3, 16, 24)).astype(np.float32)
    ty, tz, th = T(logp_y, True), T(logp_z, True), T(xh, True)
    res = rd_loss({"logp_y": ty, "logp_z": tz, "x_hat": th}, T(x), 0.01)
    res["loss"].backward()
    kw = {k: np.float64(v) for k, v in res.items() if isinstance(v, float)}
    kw["loss"] = np.float64(res["loss"].item())
    save("op_rd_loss.npz", logp_y=logp_y, logp_z=logp_z, x=x, x_hat=xh, lambda_rd=0.01,
         mse_per_image=res["mse_per_image"].numpy(), psnr_per_image=res["psnr_per_image"].numpy(),
         dlogp_y=ty.grad.numpy(), dlogp_z=tz.grad.numpy(), dx_hat=th.grad.numpy(), **kw)


def gen_vision_rd_loss():
    """RateDistortionLoss.py:52-121 (the scalable model's loss; the function itself imports and runs although the
    model it was written for does not): with and without the vision term.  V / frozen_activation are
    arbitrary user modules: a 3x3 stride-2 convolution with recipe weights (stored in the fixture) and a Tanh."""
    from RateDistortionLoss import vision_rd_loss
    B = 3
    r = np.random.RandomState(71)
    logp_y1 = np.log(r.uniform(1e-9, 1.0, (B, 4, 4, 6))).astype(np.float32)
    logp_y2 = np.log(r.uniform(1e-6, 1.0, (B, 4, 4, 6))).astype(np.float32)
    logp_z = np.log(r.uniform(1e-3, 1.0, (B, 8, 1, 2))).astype(np.float32)
    x = r.rand(B, 3, 16, 24).astype(np.float32)
    xh = (x + 0.05 * r.randn(B, 3, 16, 24)).astype(np.float32)
    ft = r.randn(B, 5, 8, 12).astype(np.float32)
    vw = (r.randn(5, 3, 3, 3) / 5.0).astype(np.float32)
    vb = (0.1 * r.randn(5)).astype(np.float32)
    V = nn.Conv2d(3, 5, 3, stride=2, padding=1)
    with torch.no_grad():
        V.weight.copy_(T(vw))
        V.bias.copy_(T(vb))
    act = nn.Tanh()
    kw = dict(logp_y1=logp_y1, logp_y2=logp_y2, logp_z=logp_z, x=x, x_hat=xh, F_tilde=ft, v_weight=vw, v_bias=vb,
              lambda_rd=0.013, gamma=0.7)
    for tag, mods in (("plain", (None, None)), ("vision", (act, V))):
        t1, t2, tz, th, tf = T(logp_y1, True), T(logp_y2, True), T(logp_z, True), T(xh, True), T(ft, True)
        for q in V.parameters():
            q.grad = None
        res = vision_rd_loss({"logp_y1": t1, "logp_y2": t2, "logp_z": tz, "x_hat": th, "F_tilde": tf}, T(x), 0.013, 0.7,
                             frozen_activation=mods[0], V=mods[1])
        res["loss"].backward()
        for k, v in res.items():
            if isinstance(v, float):
                kw[f"{tag}.{k}"] = np.float64(v)
            elif torch.is_tensor(v) and k != "loss":
                kw[f"{tag}.{k}"] = v.numpy()
        kw[f"{tag}.loss"] = np.float64(res["loss"].item())
        kw[f"{tag}.dlogp_y1"], kw[f"{tag}.dlogp_y2"] = t1.grad.numpy(), t2.grad.numpy()
        kw[f"{tag}.dlogp_z"], kw[f"{tag}.dx_hat"] = tz.grad.numpy(), th.grad.numpy()
        if mods[1] is not None:
            kw[f"{tag}.dF_tilde"] = tf.grad.numpy()
            kw[f"{tag}.dv_weight"] = V.weight.grad.numpy().copy()
    save("op_vision_rd_loss.npz", **kw)


def gen_masked_conv():
    from ContextModels import ContextModel
    M, B, h, w = 4, 2, 6, 5
    m = ContextModel(M)
    ks = load_recipe_state(m, seed=71)
    xin = (R.make_noise((B, M, h, w), 72) * 6 - 3).astype(np.float32)
    x = T(xin, True)
    y = m(x)
    cot = R.make_noise(tuple(y.shape), 73) - 0.5
    (y * T(cot)).sum().backward()
    save("op_masked_conv.npz", M=M, keys_shapes=ks_json(ks), x=xin, y=y.detach().numpy(),
         weight_after=m.masked.weight.detach().numpy(), mask=m.masked.mask.numpy(),
         dx=x.grad.numpy(), dweight=m.masked.weight.grad.numpy(), dbias=m.masked.bias.grad.numpy(),
         seed_state=71, seed_cot=73)


# --- whole model, GDN-unpinned --------------------------------------------------------------------
def gen_model(kind, M, K, B, H, W, seed, lam=0.01):
    import Models
    cls = Models.JointAutoregressiveHierarchical if kind == "5x5" else Models.HierarchicalMixtureResidual
    m = cls(M, K)
    ks = load_recipe_state(m, seed=seed)
    from RateDistortionLoss import rd_loss
    x = R.make_image(B, H, W, seed + 1)
    kw = {"keys_shapes": ks_json(ks)}
    # eval mode (deterministic)
    m.eval()
    with torch.no_grad():
        out = m(T(x), training=False)
        res = rd_loss(out, T(x), lam)
    for k in ("x_hat", "y", "z", "y_in", "z_in", "p_y", "p_z", "logp_y", "logp_z"):
        kw["eval." + k] = out[k].numpy()
    for k in ("mu", "sigma", "weights", "mus", "sigmas"):
        if k in out:
            kw["eval." + k] = out[k].numpy()
    for k, v in res.items():
        if isinstance(v, float):
            kw["eval.loss." + k] = np.float64(v)
    kw["eval.loss.loss"] = np.float64(res["loss"].item())
    # train mode with injected noise (torch.rand_like patched; reference draws z then y)
    m.train()
    # load again: the masked-conv forward zeroed taps in place -- same values either way
    zshape, yshape = tuple(out["z"].shape), tuple(out["y"].shape)
    uz, uy = R.make_noise(zshape, seed + 2), R.make_noise(yshape, seed + 3)
    queue = [T(uz), T(uy)]
    orig = torch.rand_like
    torch.rand_like = lambda t, *a, **k: queue.pop(0)
    try:
        out = m(T(x))
    finally:
        torch.rand_like = orig
    assert not queue
    res = rd_loss(out, T(x), lam)
    res["loss"].backward()
    for k in ("x_hat", "y", "z", "y_in", "z_in", "p_y", "p_z", "logp_y", "logp_z"):
        kw["train." + k] = out[k].detach().numpy()
    for k, v in res.items():
        if isinstance(v, float):
            kw["train.loss." + k] = np.float64(v)
    kw["train.loss.loss"] = np.float64(res["loss"].item())
    for k, p in m.named_parameters():
        g = p.grad.numpy()
        kw["grad." + k] = subsample(g)
        kw["gradnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
    save(f"model_{'jah' if kind == '5x5' else 'hmr'}_M{M}_K{K}.npz", kind=kind, M=M, K=K, B=B, H=H,
         W=W, seed=seed, lambda_rd=lam, gdn_pinned=False, **kw)


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_entropy_parameters(1)
    gen_entropy_parameters(3)
    gen_factorized()
    gen_gaussian()
    gen_rd_loss()
    gen_vision_rd_loss()
    gen_masked_conv()
    _register_gdn_standin()
    gen_model("5x5", 8, 1, 2, 64, 64, seed=101)
    gen_model("5x5", 8, 3, 2, 128, 64, seed=111)
    gen_model("3x3", 8, 3, 1, 64, 64, seed=121)


if __name__ == "__main__":
    main()
