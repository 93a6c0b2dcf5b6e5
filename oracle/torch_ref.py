"""Second CPU restatement of the hot path, written with plain torch ops (torch.nn.functional on
CPU, autograd for the backward): the closest thing to "the reference's PyTorch CPU path" that can
run where the reference itself cannot travel.  TEST INFRASTRUCTURE ONLY (same rules as
lic_oracle.c): used by tests/ as a cross-check of the C oracle and of the HIP path at sizes where
oneDNN is much faster than the plain-C loops, and by bench.py's cpu_baseline leg.

Restates (paths under /root/reference): Models.py:49-106,148-205; Components.py:6-122;
Layers.py:18-119; ContextModels.py:9-20; ParametersModels.py:20-64; EntropyModels.py:29-31,
88-151,192-233; utils.py:6-8; RateDistortionLoss.py:5-49; GDN per SURVEY.md Appendix B
(third-party compressai: parity unpinned).  Parameters: a state dict with the reference's keys.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

PEDESTAL = float(2.0 ** -36)


class _LowerBound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = bound
        return torch.clamp_min(x, bound)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ((x >= ctx.bound) | (g < 0)).to(g.dtype) * g, None


def _gdn(x, P, pre, inverse):
    C = x.size(1)
    beta = _LowerBound.apply(P[pre + ".beta"], (1e-6 + PEDESTAL) ** 0.5) ** 2 - PEDESTAL
    gamma = _LowerBound.apply(P[pre + ".gamma"], PEDESTAL ** 0.5) ** 2 - PEDESTAL
    norm = F.conv2d(x * x, gamma.reshape(C, C, 1, 1), beta)
    return x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))


def _conv(x, P, pre, s, p):
    return F.conv2d(x, P[pre + ".weight"], P[pre + ".bias"], stride=s, padding=p)


def _convT(x, P, pre, s, p, op):
    return F.conv_transpose2d(x, P[pre + ".weight"], P[pre + ".bias"], stride=s, padding=p, output_padding=op)


def _lrelu(x):
    return F.leaky_relu(x, 0.01)


def _rb_stride(x, P, pre):
    out = _gdn(_conv(_lrelu(_conv(x, P, pre + ".conv1", 2, 1)), P, pre + ".conv2", 1, 1), P, pre + ".gdn", False)
    idn = _conv(x, P, pre + ".skip", 2, 0) if (pre + ".skip.weight") in P else x
    return out + idn


def _rb(x, P, pre):
    out = _lrelu(_conv(_lrelu(_conv(x, P, pre + ".conv1", 1, 1)), P, pre + ".conv2", 1, 1))
    idn = _conv(x, P, pre + ".skip", 1, 0) if (pre + ".skip.weight") in P else x
    return out + idn


def _rb_up(x, P, pre):
    out = _lrelu(_convT(x, P, pre + ".subpel_conv.deconv", 2, 1, 1))
    out = _gdn(_conv(out, P, pre + ".conv", 1, 1), P, pre + ".igdn", True)
    return out + _convT(x, P, pre + ".upsample.deconv", 2, 1, 1)


def encoder(x, P, kind):
    p = "encoder.net."
    if kind == "5x5":
        h = _gdn(_conv(x, P, p + "0", 2, 2), P, p + "1", False)
        h = _gdn(_conv(h, P, p + "2", 2, 2), P, p + "3", False)
        h = _gdn(_conv(h, P, p + "4", 2, 2), P, p + "5", False)
        return _conv(h, P, p + "6", 2, 2)
    h = _rb(_rb_stride(x, P, p + "0"), P, p + "1")
    h = _rb(_rb_stride(h, P, p + "2"), P, p + "3")
    h = _rb(_rb_stride(h, P, p + "4"), P, p + "5")
    return _conv(h, P, p + "6", 2, 1)


def decoder(y, P, kind):
    p = "decoder.net."
    if kind == "5x5":
        h = _gdn(_convT(y, P, p + "0", 2, 2, 1), P, p + "1", True)
        h = _gdn(_convT(h, P, p + "2", 2, 2, 1), P, p + "3", True)
        h = _gdn(_convT(h, P, p + "4", 2, 2, 1), P, p + "5", True)
        return _convT(h, P, p + "6", 2, 2, 1)
    h = _rb_up(_rb(y, P, p + "0"), P, p + "1")
    h = _rb_up(_rb(h, P, p + "2"), P, p + "3")
    h = _rb_up(_rb(h, P, p + "4"), P, p + "5")
    return _convT(_rb(h, P, p + "6"), P, p + "7.deconv", 2, 1, 1)


def hyper_encoder(y, P, kind):
    p = "hyper_encoder.net."
    if kind == "5x5":
        h = _lrelu(_conv(y, P, p + "0", 1, 1))
        h = _lrelu(_conv(h, P, p + "2", 2, 2))
        return _conv(h, P, p + "4", 2, 2)
    h = _lrelu(_conv(y, P, p + "0", 1, 1))
    h = _lrelu(_conv(h, P, p + "2", 1, 1))
    h = _lrelu(_conv(h, P, p + "4", 2, 1))
    h = _lrelu(_conv(h, P, p + "6", 1, 1))
    return _conv(h, P, p + "8", 2, 1)


def hyper_decoder(z, P, kind):
    p = "hyper_decoder.net."
    if kind == "5x5":
        h = _lrelu(_convT(z, P, p + "0", 2, 2, 1))
        h = _lrelu(_convT(h, P, p + "2", 2, 2, 1))
        return _conv(h, P, p + "4", 1, 1)
    h = _lrelu(_conv(z, P, p + "0", 1, 1))
    h = _lrelu(_convT(h, P, p + "2.deconv", 2, 1, 1))
    h = _lrelu(_conv(h, P, p + "4", 1, 1))
    h = _lrelu(_convT(h, P, p + "6.deconv", 2, 1, 1))
    return _conv(h, P, p + "8", 1, 1)


def _factorized(x, P):
    """EntropyModels.py:88-151 (sign trick with detached sign)."""
    pre = "factorized_entropy_model."
    C = x.size(1)
    flat = x.transpose(0, 1).reshape(C, 1, -1)

    def logits(v):
        for i in range(4):
            v = torch.matmul(F.softplus(P[pre + f"matrices.{i}"]), v) + P[pre + f"biases.{i}"]
            if i < 3:
                v = v + torch.tanh(P[pre + f"factors.{i}"]) * torch.tanh(v)
        return v
    lower, upper = logits(flat - 0.5), logits(flat + 0.5)
    s = -torch.sign(lower + upper).detach()
    pmf = torch.abs(torch.sigmoid(s * upper) - torch.sigmoid(s * lower))
    return pmf.reshape(C, x.size(0), *x.shape[2:]).transpose(0, 1)


def _gcdf(t):
    return 0.5 * (1.0 + torch.erf(t / math.sqrt(2.0)))


def forward(P: Dict[str, torch.Tensor], x: torch.Tensor, M: int, K: int, kind: str = "5x5",
            training: bool = True, noise: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Returns the reference's 13-key dict (Models.py:92-106)."""
    y = encoder(x, P, kind)
    z = hyper_encoder(y, P, kind)
    if training:
        uz, uy = noise if noise is not None else (torch.rand_like(z), torch.rand_like(y))
        z_in, y_in = z + (uz - 0.5), y + (uy - 0.5)
    else:
        z_in, y_in = torch.round(z), torch.round(y)
    psi = hyper_decoder(z_in, P, kind)
    w = P["context_model.masked.weight"]
    mask = torch.ones_like(w)
    mask[:, :, 2, 2:] = 0
    mask[:, :, 3:] = 0
    with torch.no_grad():
        w.mul_(mask)  # ContextModels.py:19 (in place, outside autograd)
    phi = F.conv2d(y_in, w, P["context_model.masked.bias"], padding=2)
    h = torch.cat([phi, psi], dim=1)
    ep = "entropy_parameters.net."
    h = _lrelu(_conv(h, P, ep + "0", 1, 0))
    h = _lrelu(_conv(h, P, ep + "2", 1, 0))
    raw = _conv(h, P, ep + "4", 1, 0)
    out = {}
    if K == 1:
        mu, sg = raw.chunk(2, dim=1)
        sg = F.softplus(sg) + 1e-6
        p_y = _gcdf((y_in + 0.5 - mu) / sg) - _gcdf((y_in - 0.5 - mu) / sg)
        out.update(mu=mu, sigma=sg)
    else:
        B, _, hh, ww = raw.shape
        wt, mus, sgs = (t.reshape(B, K, M, hh, ww) for t in raw.chunk(3, dim=1))
        wt = F.softmax(wt, dim=1)
        sgs = F.softplus(sgs) + 1e-6
        xe = y_in.unsqueeze(1)
        p_y = (wt * (_gcdf((xe + 0.5 - mus) / sgs) - _gcdf((xe - 0.5 - mus) / sgs))).sum(dim=1)
        out.update(weights=wt, mus=mus, sigmas=sgs)
    p_y = p_y.clamp_min(1e-9)
    p_z = _factorized(z_in, P).clamp_min(1e-9)
    out.update(x_hat=decoder(y_in, P, kind), y=y, y_in=y_in, z=z, z_in=z_in, p_z=p_z,
               logp_z=torch.log(p_z), p_y=p_y, logp_y=torch.log(p_y), training=training)
    return out


def rd_loss(out, x, lambda_rd):
    """RateDistortionLoss.py:5-49 (tensor results, no .item())."""
    npix = x.size(2) * x.size(3)
    bits_y = -out["logp_y"].sum(dim=(1, 2, 3)) / math.log(2.0)
    bits_z = -out["logp_z"].sum(dim=(1, 2, 3)) / math.log(2.0)
    bpp_y, bpp_z = (bits_y / npix).mean(), (bits_z / npix).mean()
    mse_img = ((out["x_hat"] - x) ** 2).mean(dim=(1, 2, 3))
    mse = mse_img.mean()
    return {"loss": bpp_y + bpp_z + lambda_rd * 255 ** 2 * mse, "bpp_y": bpp_y, "bpp_z": bpp_z,
            "bpp_total": bpp_y + bpp_z, "mse": mse, "psnr": -10 * torch.log10(mse + 1e-8),
            "bits_y": bits_y.mean(), "bits_z": bits_z.mean(), "mse_per_image": mse_img}


def step(state: Dict[str, "object"], x, M, K, kind="5x5", noise=None, lambda_rd=0.01):
    """One forward + rd_loss + backward on CPU; `state` values may be numpy arrays or tensors.
    Returns (out, loss dict of floats, grads dict of numpy arrays)."""
    import numpy as np
    P = {}
    for k, v in state.items():
        t = torch.as_tensor(np.asarray(v)).clone() if not torch.is_tensor(v) else v.detach().clone().cpu()
        if t.is_floating_point() and k.split(".")[-1] not in ("pedestal", "bound", "mask"):
            t.requires_grad_(True)
        P[k] = t
    xt = torch.as_tensor(np.asarray(x)) if not torch.is_tensor(x) else x.cpu()
    nz = None if noise is None else tuple(torch.as_tensor(np.asarray(n)) for n in noise)
    out = forward(P, xt, M, K, kind, True, nz)
    res = rd_loss(out, xt, lambda_rd)
    res["loss"].backward()
    grads = {k: v.grad.numpy() for k, v in P.items() if v.requires_grad and v.grad is not None}
    loss = {k: float(v.detach()) for k, v in res.items() if v.dim() == 0}
    return {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}, loss, grads


# ----------------------------------------------------------------------------------------------
# ScalableImageCoding / LatentSpaceTransform / vision_rd_loss (Models.py:208-338, Components.py:125-153,
# RateDistortionLoss.py:52-121) with the three repairs the product states (models.ScalableImageCoding): the
# reference's versions cannot execute, so there is nothing upstream to pin these against.
# ----------------------------------------------------------------------------------------------
def _rb_up_f(x, P, pre, up):
    out = _lrelu(_convT(x, P, pre + ".subpel_conv.deconv", up, 1, up - 1))
    out = _gdn(_conv(out, P, pre + ".conv", 1, 1), P, pre + ".igdn", True)
    return out + _convT(x, P, pre + ".upsample.deconv", up, 1, up - 1)


def latent_space_transform(x, P, pre="LST", ups=(2, 1, 1, 1)):
    for i in range(3):
        x = _rb_up_f(_rb(x, P, f"{pre}.RB{i + 1}"), P, f"{pre}.URB{i + 1}", ups[i])
    return _conv(_rb(x, P, pre + ".RB4"), P, pre + ".conv", 1, 1)


def _masked(yk, P, pre):
    w = P[pre + ".masked.weight"]
    mask = torch.ones_like(w)
    mask[:, :, 2, 2:] = 0
    mask[:, :, 3:] = 0
    with torch.no_grad():
        w.mul_(mask)
    return F.conv2d(yk, w, P[pre + ".masked.bias"], padding=2)


def _conditional(yk, raw, Mk, K):
    if K == 1:
        mu, sg = raw.chunk(2, dim=1)
        sg = F.softplus(sg) + 1e-6
        return (_gcdf((yk + 0.5 - mu) / sg) - _gcdf((yk - 0.5 - mu) / sg)).clamp_min(1e-9), (mu, sg)
    B, _, hh, ww = raw.shape
    wt, mus, sgs = (t.reshape(B, K, Mk, hh, ww) for t in raw.chunk(3, dim=1))
    wt, sgs = F.softmax(wt, dim=1), F.softplus(sgs) + 1e-6
    xe = yk.unsqueeze(1)
    p = (wt * (_gcdf((xe + 0.5 - mus) / sgs) - _gcdf((xe - 0.5 - mus) / sgs))).sum(dim=1)
    return p.clamp_min(1e-9), (wt, mus, sgs)


def forward_scalable(P, x, M, M1, K, training=True, noise=None):
    y = encoder(x, P, "5x5")
    z = hyper_encoder(y, P, "5x5")
    if training:
        uz, uy = noise if noise is not None else (torch.rand_like(z), torch.rand_like(y))
        z_in, y_in = z + (uz - 0.5), y + (uy - 0.5)
    else:
        z_in, y_in = torch.round(z), torch.round(y)
    y1, y2 = torch.split(y_in, [M1, M - M1], dim=1)
    psi = hyper_decoder(z_in, P, "5x5")
    out = {"y": y, "y_in": y_in, "y1": y1, "y2": y2, "z": z, "z_in": z_in, "training": training}
    for tag, yk, Mk in (("1", y1, M1), ("2", y2, M - M1)):
        h = torch.cat([_masked(yk, P, "context_model_" + tag), psi], dim=1)
        ep = f"entropy_parameters_{tag}.net."
        h = _lrelu(_conv(h, P, ep + "0", 1, 0))
        h = _lrelu(_conv(h, P, ep + "2", 1, 0))
        p, par = _conditional(yk, _conv(h, P, ep + "4", 1, 0), Mk, K)
        out["p_y" + tag], out["logp_y" + tag] = p, torch.log(p)
        for n, v in zip(("mu", "sigma") if K == 1 else ("weights", "mus", "sigmas"), par):
            out[n + tag] = v
    p_z = _factorized(z_in, P).clamp_min(1e-9)
    out.update(p_z=p_z, logp_z=torch.log(p_z), x_hat=decoder(y_in, P, "5x5"), F_tilde=latent_space_transform(y1, P))
    return out


def vision_rd_loss(out, x, lambda_rd, gamma, frozen_activation=None, V=None):
    npix = x.size(2) * x.size(3)
    bits = {k: -out["logp_" + k].sum(dim=(1, 2, 3)) / math.log(2.0) for k in ("y1", "y2", "z")}
    bpp = {k: (v / npix).mean() for k, v in bits.items()}
    rec_img = ((out["x_hat"] - x) ** 2).mean(dim=(1, 2, 3))
    rec = rec_img.mean()
    mse, vis = rec, None
    if frozen_activation is not None and V is not None:
        vis = ((frozen_activation(out["F_tilde"]) - V(out["x_hat"])) ** 2).mean(dim=(1, 2, 3)).mean()
        mse = rec + gamma * vis
    res = {"loss": bpp["y1"] + bpp["y2"] + bpp["z"] + lambda_rd * mse, "bpp_y1": bpp["y1"], "bpp_y2": bpp["y2"],
           "bpp_z": bpp["z"], "bpp_total": bpp["y1"] + bpp["y2"] + bpp["z"], "mse": mse, "reconstruction_mse": rec,
           "psnr": -10 * torch.log10(rec + 1e-8), "bits_y1": bits["y1"].mean(), "bits_y2": bits["y2"].mean(),
           "bits_z": bits["z"].mean()}
    if vis is not None:
        res["vision_mse"] = vis
    return res


def step_scalable(state, x, M, M1, K, noise, lambda_rd, gamma, frozen_activation=None, V=None):
    import numpy as np
    P = {}
    for k, v in state.items():
        t = torch.as_tensor(np.asarray(v)).clone()
        if t.is_floating_point() and k.split(".")[-1] not in ("pedestal", "bound", "mask"):
            t.requires_grad_(True)
        P[k] = t
    xt = torch.as_tensor(np.asarray(x))
    out = forward_scalable(P, xt, M, M1, K, True, tuple(torch.as_tensor(np.asarray(n)) for n in noise))
    res = vision_rd_loss(out, xt, lambda_rd, gamma, frozen_activation, V)
    res["loss"].backward()
    grads = {k: v.grad.numpy() for k, v in P.items() if v.requires_grad and v.grad is not None}
    return ({k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()},
            {k: float(v.detach()) for k, v in res.items()}, grads)


# ----------------------------------------------------------------------------------------------
# single layers (Components.py:12,41 at their real sizes: tests/test_gpu_variants.py)
# ----------------------------------------------------------------------------------------------
def _np(a):
    import numpy as np
    return torch.as_tensor(np.ascontiguousarray(a))


def conv2d_step(x, w, b, dy, stride, pad):
    """nn.Conv2d forward + the three gradients for the output gradient `dy`; numpy in, numpy out."""
    xt, wt, bt = _np(x).requires_grad_(True), _np(w).requires_grad_(True), _np(b).requires_grad_(True)
    y = F.conv2d(xt, wt, bt, stride=stride, padding=pad)
    y.backward(_np(dy))
    return y.detach().numpy(), xt.grad.numpy(), wt.grad.numpy(), bt.grad.numpy()


def conv_transpose2d(x, w, b, stride, pad, out_pad):
    with torch.no_grad():
        return F.conv_transpose2d(_np(x), _np(w), None if b is None else _np(b), stride=stride, padding=pad,
                                  output_padding=out_pad).numpy()


def conv_transpose2d_step(x, w, b, dy, stride, pad, out_pad):
    """nn.ConvTranspose2d forward + the three gradients; numpy in, numpy out."""
    xt, wt, bt = _np(x).requires_grad_(True), _np(w).requires_grad_(True), _np(b).requires_grad_(True)
    y = F.conv_transpose2d(xt, wt, bt, stride=stride, padding=pad, output_padding=out_pad)
    y.backward(_np(dy))
    return y.detach().numpy(), xt.grad.numpy(), wt.grad.numpy(), bt.grad.numpy()


# ----------------------------------------------------------------------------------------------
# MS-SSIM.  The reference calls `pytorch_msssim.ms_ssim(recon, orig, data_range=1.0,
# size_average=True)` (Evaluator.py:7,38,45; requirements.txt:5 pins pytorch-msssim==0.2.1, which is
# not installable offline => PARITY UNPINNED).  Restated from the package's published algorithm.
# ----------------------------------------------------------------------------------------------
def _msssim_window(size=11, sigma=1.5):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _msssim_filter(x, g):
    """separable VALID Gaussian filtering per channel, H first then W (as the package loops)"""
    C = x.shape[1]
    k = g.numel()
    if x.shape[2] >= k:
        x = F.conv2d(x, g.view(1, 1, k, 1).repeat(C, 1, 1, 1), groups=C)
    if x.shape[3] >= k:
        x = F.conv2d(x, g.view(1, 1, 1, k).repeat(C, 1, 1, 1), groups=C)
    return x


def _ssim_terms(X, Y, data_range, g, K=(0.01, 0.03)):
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _msssim_filter(X, g), _msssim_filter(Y, g)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _msssim_filter(X * X, g) - mu1_sq
    s2 = _msssim_filter(Y * Y, g) - mu2_sq
    s12 = _msssim_filter(X * Y, g) - mu1_mu2
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(X, Y, data_range=255.0, size_average=True):
    X, Y = X.float(), Y.float()
    if min(X.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("Image size should be larger than 160 due to the 4 downsamplings in ms-ssim")
    weights = torch.tensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])
    g = _msssim_window()
    mcs = []
    for i in range(5):
        ssim_c, cs = _ssim_terms(X, Y, data_range, g)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    stack = torch.stack(mcs + [torch.relu(ssim_c)], dim=0)  # (level, batch, channel)
    val = torch.prod(stack ** weights.view(-1, 1, 1), dim=0)
    return val.mean() if size_average else val.mean(1)
