"""BASELINE.json configs[0] (SURVEY.md 8(d) cfg 1): "Factorised-prior model, capacity 128, 4x256x256 random
images, CPU reference path via Models.py (plumbing, no GPU)".

The reference has no factorised-prior model class; SURVEY D1 maps the config to (a) the composition
`Encoder5x5(128) -> FactorizedEntropyBottleneck(128) on y -> Decoder5x5(128)` (Components.py:6-18,35-47;
EntropyModels.py:49-151) and (b) `JointAutoregressiveHierarchical(128, K=1)` (Models.py:49-106), both at
4x3x256x256 on the CPU.  CPU part: the two independent CPU restatements -- the plain-C oracle and the
torch-op restatement (torch CPU ops are the arithmetic of the reference's own CPU path) -- must agree at
that size within the north star's 1e-4.  GPU part (-m gpu): one image of each through the HIP path
against the oracle.
"""
import math

import numpy as np
import pytest
import torch

import golden_recipe as R

M, B, H, W, LAM = 128, 4, 256, 256, 0.01


def _state(seed=11):
    import neural_image_compression_amd as nic
    model = nic.JointAutoregressiveHierarchical(M, 1)
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    return R.make_state(ks, seed)


def _rel(a, b, rtol=1e-4, atol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    assert (err <= atol + rtol * np.abs(b)).all(), f"max err {err.max():.3e} (|ref| max {np.abs(b).max():.3e})"


def _factorised_prior_c_oracle(st, x, uy, backward):
    """the composition through oracle/lic_oracle.c; returns (out, loss terms, parameter grads)"""
    from oracle import oracle as O
    t = O.Tape(st)
    xv = O.V(np.ascontiguousarray(x, np.float32))
    y = O.encoder(t, xv, "5x5")
    y_in = O.V(O.quantize(y.d, uy, True))

    def bw_q():
        if y_in.g is not None:
            y.acc(y_in.g)
    t.fns.append(bw_q)
    pre = "factorized_entropy_model."
    keys = ([pre + f"matrices.{i}" for i in range(4)], [pre + f"biases.{i}" for i in range(4)],
            [pre + f"factors.{i}" for i in range(3)])
    packed = O.fe_pack(*[[t.params[k] for k in ks] for ks in keys])
    p_y, logp_y = O.factorized_fwd(y_in.d, packed)
    x_hat = O.decoder(t, y_in, "5x5")
    npix = x.shape[2] * x.shape[3]
    bpp = float((-logp_y.astype(np.float64).sum(axis=(1, 2, 3)) / math.log(2.0) / npix).mean())
    mse = float(((x_hat.d.astype(np.float64) - x) ** 2).mean(axis=(1, 2, 3)).mean())
    grads = None
    if backward:  # d(bpp + lam*255^2*mse)
        Bn = x.shape[0]
        dlog = np.full_like(logp_y, -1.0 / (math.log(2.0) * npix * Bn))
        dxh = (2.0 * LAM * 255 ** 2 / (Bn * 3 * npix) * (x_hat.d - x)).astype(np.float32)
        dx, dpk = O.factorized_bwd(y_in.d, packed, None, dlog)
        y_in.acc(dx)
        mats, bs, fs = O.fe_unpack(dpk)
        for k, g in zip(keys[0] + keys[1] + keys[2], mats + bs + fs):
            t.pacc(k, g)
        x_hat.acc(dxh)
        t.backward()
        grads = t.pgrad
    return {"y": y.d, "y_in": y_in.d, "p_y": p_y, "logp_y": logp_y, "x_hat": x_hat.d}, {"bpp": bpp, "mse": mse}, grads


def _factorised_prior_torch(st, x, uy, backward):
    from oracle import torch_ref as TR
    P = {}
    for k, v in st.items():
        tns = torch.as_tensor(np.asarray(v)).clone()
        if backward and tns.is_floating_point() and k.split(".")[-1] not in ("pedestal", "bound", "mask"):
            tns.requires_grad_(True)
        P[k] = tns
    xt = torch.as_tensor(x)
    with torch.set_grad_enabled(backward):
        y = TR.encoder(xt, P, "5x5")
        y_in = y + (torch.as_tensor(uy) - 0.5)
        p_y = TR._factorized(y_in, P).clamp_min(1e-9)
        logp_y = torch.log(p_y)
        x_hat = TR.decoder(y_in, P, "5x5")
        npix = x.shape[2] * x.shape[3]
        bpp = (-logp_y.sum(dim=(1, 2, 3)) / math.log(2.0) / npix).mean()
        mse = ((x_hat - xt) ** 2).mean(dim=(1, 2, 3)).mean()
        grads = None
        if backward:
            (bpp + LAM * 255 ** 2 * mse).backward()
            grads = {k: v.grad.numpy() for k, v in P.items() if v.requires_grad and v.grad is not None}
    out = {k: v.detach().numpy() for k, v in dict(y=y, y_in=y_in, p_y=p_y, logp_y=logp_y, x_hat=x_hat).items()}
    return out, {"bpp": float(bpp.detach()), "mse": float(mse.detach())}, grads


def test_cfg1_factorised_prior_composition_cpu():
    """4x3x256x256 through Encoder5x5(128) -> FactorizedEntropyBottleneck(128)(y) -> Decoder5x5(128):
    C oracle == torch-op restatement (forward, bpp, mse), and for the first image the backward too."""
    st = _state()
    x = R.make_image(B, H, W, 12)
    uy = R.make_noise((B, M, H // 16, W // 16), 13)
    c_out, c_loss, _ = _factorised_prior_c_oracle(dict(st), x, uy, False)
    t_out, t_loss, _ = _factorised_prior_torch(st, x, uy, False)
    assert c_out["x_hat"].shape == (B, 3, H, W) and c_out["y"].shape == (B, M, H // 16, W // 16)
    for k in ("y", "x_hat", "logp_y"):
        _rel(c_out[k], t_out[k], 1e-4, 1e-4 if k != "logp_y" else 2e-5)
    for k in ("bpp", "mse"):
        assert abs(c_loss[k] - t_loss[k]) <= 1e-4 * abs(t_loss[k]), (k, c_loss[k], t_loss[k])
    assert 0.0 < c_loss["bpp"] < 64.0 and c_loss["mse"] > 0.0
    _, _, c_g = _factorised_prior_c_oracle(dict(st), x[:1], uy[:1], True)
    _, _, t_g = _factorised_prior_torch(st, x[:1], uy[:1], True)
    assert set(t_g) <= set(c_g)
    for k, ref in t_g.items():
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(c_g[k] - ref).max() <= 3e-4 * scale + 3e-7, (k, np.abs(c_g[k] - ref).max(), scale)


def test_cfg1_jah128_cpu():
    """JointAutoregressiveHierarchical(128, K=1) at 4x3x256x256: C oracle == torch-op restatement, forward +
    rd_loss (lambda 0.01) + every parameter gradient."""
    from oracle import oracle as O
    from oracle import torch_ref as TR
    st = _state(21)
    x = R.make_image(B, H, W, 22)
    noise = (R.make_noise((B, M, H // 64, W // 64), 23), R.make_noise((B, M, H // 16, W // 16), 24))
    o_out, o_loss, o_g = O.model_forward(dict(st), x, M, 1, "5x5", training=True, noise=noise, lambda_rd=LAM,
                                         backward=True)
    t_out, t_loss, t_g = TR.step(st, x, M, 1, "5x5", noise, LAM)
    assert o_out["x_hat"].shape == (B, 3, H, W) and o_out["z"].shape == (B, M, H // 64, W // 64)
    for k in ("y", "z", "x_hat", "mu", "sigma"):
        _rel(o_out[k], t_out[k])
    for k in ("bpp_y", "bpp_z", "bpp_total", "mse", "psnr", "loss"):
        assert abs(o_loss[k] - t_loss[k]) <= 1e-4 * abs(t_loss[k]), (k, o_loss[k], t_loss[k])
    for k, ref in t_g.items():
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(o_g[k] - ref).max() <= 5e-4 * scale + 3e-7, (k, np.abs(o_g[k] - ref).max(), scale)


@pytest.mark.gpu
def test_cfg1_one_image_of_each_through_the_hip_path():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    from neural_image_compression_amd import functional as F_
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    st = _state(31)
    x = R.make_image(1, H, W, 32)
    uz, uy = R.make_noise((1, M, H // 64, W // 64), 33), R.make_noise((1, M, H // 16, W // 16), 34)
    model = nic.JointAutoregressiveHierarchical(M, 1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(dev)
    tx = torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last)
    # (a) the factorised-prior composition from the model's own sub-modules
    y = model.encoder(tx)
    y_in = F_.quantize(y, torch.from_numpy(uy).to(dev), True)
    p_y, logp_y = model.factorized_entropy_model.likelihood_and_log(y_in)
    x_hat = model.decoder(y_in)
    c_out, c_loss, _ = _factorised_prior_c_oracle(dict(st), x, uy, False)
    for k, v in (("y", y), ("logp_y", logp_y), ("x_hat", x_hat)):
        _rel(v.detach().cpu().numpy(), c_out[k], 1e-4, 1e-4 if k != "logp_y" else 2e-5)
    npix = H * W
    bpp = float((-logp_y.double().sum() / math.log(2.0) / npix))
    assert abs(bpp - c_loss["bpp"]) <= 1e-4 * c_loss["bpp"]
    # (b) JAH(128, 1)
    out = model(tx, noise=(torch.from_numpy(uz).to(dev), torch.from_numpy(uy).to(dev)))
    res = nic.rd_loss(out, tx, LAM)
    res["loss"].backward()
    o_out, o_loss, o_g = O.model_forward(dict(st), x, M, 1, "5x5", training=True, noise=(uz, uy), lambda_rd=LAM,
                                         backward=True)
    for k in ("y", "z", "x_hat"):
        _rel(out[k].detach().cpu().numpy(), o_out[k])
    for k in ("bpp_y", "bpp_z", "mse", "psnr"):
        assert abs(res[k] - o_loss[k]) <= 1e-4 * abs(o_loss[k]), (k, res[k], o_loss[k])
    for name, p in model.named_parameters():
        ref = o_g[name]
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(p.grad.detach().cpu().numpy() - ref).max() <= 5e-4 * scale + 3e-7, name
