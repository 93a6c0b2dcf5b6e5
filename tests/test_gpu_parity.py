"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden fixtures.  fp32 tolerance of the north star: 1e-4 relative (stated per assertion).
Run on the MI355X box:  python -m pytest tests -m gpu -x -q"""
import json
import os

import numpy as np
import pytest
import torch

import golden_recipe as R

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # BASELINE.json north_star: "within 1e-4 relative fp32"


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    from neural_image_compression_amd import functional as F_
    from oracle import oracle as O
    from neural_image_compression_amd import _lib
    _lib.load()  # must be the in-tree HIP extension; raises if missing
    return nic, F_, O, torch.device("cuda:0")


def dev_nchw(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if t.dim() == 4:
        t = t.contiguous(memory_format=torch.channels_last)
    if grad:
        t.requires_grad_(True)
    return t


def host(t):
    return t.detach().cpu().contiguous().numpy()


def close(a, b, rtol=RTOL, atol=1e-6, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} bad, max err {err.max():.3e} (|ref| max {np.abs(b).max():.3e})"


def close_norm(a, b, rtol=RTOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= rtol * scale, f"{what}: {np.abs(a - b).max():.3e} vs scale {scale:.3e}"


# ---------------------------------------------------------------------------------------------
# convolution family
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # k, stride, pad, Cin, Cout, H, W, B, leaky
    (5, 2, 2, 16, 16, 16, 16, 2, False),     # encoder 5x5 s2
    (5, 2, 2, 32, 48, 18, 10, 3, False),     # ragged spatial, non-square
    (3, 1, 1, 16, 24, 8, 8, 2, True),        # hyper 3x3 + fused LeakyReLU
    (1, 1, 0, 64, 640, 4, 6, 2, True),       # entropy-parameter 1x1
    (1, 1, 0, 640, 72, 4, 6, 2, False),      # 640 -> 3KM
    (3, 2, 1, 8, 8, 9, 11, 1, False),        # odd sizes, Cin < chunk
    (1, 2, 0, 16, 16, 8, 8, 2, False),       # strided 1x1 skip
    (5, 1, 2, 12, 20, 6, 6, 2, False),       # Cout not a multiple of 32
    (3, 1, 1, 6, 10, 5, 5, 1, False),        # scalar (non-vector) path: C % 4 != 0
]


@pytest.mark.parametrize("k,s,p,ci,co,H,W,B,leaky", CONV_CASES)
def test_conv2d(env, k, s, p, ci, co, H, W, B, leaky):
    nic, F_, O, dev = env
    r = np.random.RandomState(ci * 100 + co)
    x = r.randn(B, ci, H, W).astype(np.float32)
    w = (r.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float32)
    b = r.randn(co).astype(np.float32)
    tx, tw, tb = dev_nchw(x, dev, True), dev_nchw(w, dev).contiguous().requires_grad_(True), dev_nchw(b, dev, True)
    ty = F_.conv2d(tx, tw, tb, s, p, leaky)
    y = O.conv2d_fwd(x, w, b, s, p)
    if leaky:
        y = O.leaky_relu_fwd(y)
    close(host(ty), y, RTOL, 1e-5, "y")
    dy = r.randn(*y.shape).astype(np.float32)
    ty.backward(dev_nchw(dy, dev))
    g = O.leaky_relu_bwd(y, dy) if leaky else dy
    dx, dw, db = O.conv2d_bwd(x, w, g, s, p)
    close_norm(host(tx.grad), dx, RTOL, "dx")
    close_norm(host(tw.grad), dw, RTOL, "dw")
    close_norm(host(tb.grad), db, RTOL, "db")


CONVT_CASES = [
    (5, 2, 2, 1, 16, 16, 4, 4, 2, False),
    (5, 2, 2, 1, 32, 48, 5, 3, 2, True),
    (3, 2, 1, 1, 16, 24, 6, 4, 2, True),
    (3, 1, 1, 0, 8, 8, 6, 6, 1, False),
    (5, 2, 2, 1, 6, 10, 3, 5, 1, False),     # scalar path
]


@pytest.mark.parametrize("k,s,p,op,ci,co,H,W,B,leaky", CONVT_CASES)
def test_conv_transpose2d(env, k, s, p, op, ci, co, H, W, B, leaky):
    nic, F_, O, dev = env
    r = np.random.RandomState(ci * 100 + co + 7)
    x = r.randn(B, ci, H, W).astype(np.float32)
    w = (r.randn(ci, co, k, k) / np.sqrt(ci * k * k)).astype(np.float32)
    b = r.randn(co).astype(np.float32)
    tx, tw, tb = dev_nchw(x, dev, True), dev_nchw(w, dev).contiguous().requires_grad_(True), dev_nchw(b, dev, True)
    ty = F_.conv_transpose2d(tx, tw, tb, s, p, op, leaky)
    y = O.convT2d_fwd(x, w, b, s, p, op)
    if leaky:
        y = O.leaky_relu_fwd(y)
    close(host(ty), y, RTOL, 1e-5, "y")
    dy = r.randn(*y.shape).astype(np.float32)
    ty.backward(dev_nchw(dy, dev))
    g = O.leaky_relu_bwd(y, dy) if leaky else dy
    dx, dw, db = O.convT2d_bwd(x, w, g, s, p, op)
    close_norm(host(tx.grad), dx, RTOL, "dx")
    close_norm(host(tw.grad), dw, RTOL, "dw")
    close_norm(host(tb.grad), db, RTOL, "db")


@pytest.mark.parametrize("k,s,p,co,H,W", [(5, 2, 2, 16, 32, 32), (3, 2, 1, 24, 16, 20), (1, 2, 0, 8, 8, 8)])
def test_image_conv_stem(env, k, s, p, co, H, W):
    """3-channel input: im2col + GEMM path (Components.py:10; Layers.py:38,43)."""
    nic, F_, O, dev = env
    r = np.random.RandomState(co)
    x = r.rand(2, 3, H, W).astype(np.float32)
    w = (r.randn(co, 3, k, k) / np.sqrt(3 * k * k)).astype(np.float32)
    b = r.randn(co).astype(np.float32)
    tx, tw, tb = dev_nchw(x, dev, True), dev_nchw(w, dev).contiguous().requires_grad_(True), dev_nchw(b, dev, True)
    ty = F_.image_conv2d(tx, tw, tb, s, p)
    y = O.conv2d_fwd(x, w, b, s, p)
    close(host(ty), y, RTOL, 1e-5, "y")
    dy = r.randn(*y.shape).astype(np.float32)
    ty.backward(dev_nchw(dy, dev))
    dx, dw, db = O.conv2d_bwd(x, w, dy, s, p)
    close_norm(host(tx.grad), dx, RTOL, "dx")
    close_norm(host(tw.grad), dw, RTOL, "dw")
    close_norm(host(tb.grad), db, RTOL, "db")


@pytest.mark.parametrize("k,s,p,op,ci,H,W", [(5, 2, 2, 1, 16, 8, 8), (3, 2, 1, 1, 24, 6, 10)])
def test_image_convT_head(env, k, s, p, op, ci, H, W):
    """3-channel output: GEMM + col2im path (Components.py:45,60)."""
    nic, F_, O, dev = env
    r = np.random.RandomState(ci)
    x = r.randn(2, ci, H, W).astype(np.float32)
    w = (r.randn(ci, 3, k, k) / np.sqrt(ci * k * k)).astype(np.float32)
    b = r.randn(3).astype(np.float32)
    tx, tw, tb = dev_nchw(x, dev, True), dev_nchw(w, dev).contiguous().requires_grad_(True), dev_nchw(b, dev, True)
    ty = F_.image_conv_transpose2d(tx, tw, tb, s, p, op)
    y = O.convT2d_fwd(x, w, b, s, p, op)
    close(host(ty), y, RTOL, 1e-5, "y")
    dy = r.randn(*y.shape).astype(np.float32)
    ty.backward(dev_nchw(dy, dev))
    dx, dw, db = O.convT2d_bwd(x, w, dy, s, p, op)
    close_norm(host(tx.grad), dx, RTOL, "dx")
    close_norm(host(tw.grad), dw, RTOL, "dw")
    close_norm(host(tb.grad), db, RTOL, "db")


@pytest.mark.parametrize("C,inverse,with_res", [(16, False, False), (16, True, False), (24, False, True),
                                                (6, True, False)])
def test_gdn(env, C, inverse, with_res):
    nic, F_, O, dev = env
    from neural_image_compression_amd.layers import GDN
    r = np.random.RandomState(C)
    x = r.randn(2, C, 6, 5).astype(np.float32)
    beta_p, gamma_p = R.make_param("g.beta", (C,), 3), R.make_param("g.gamma", (C, C), 3)
    m = GDN(C, inverse=inverse).to(dev)
    with torch.no_grad():
        m.beta.copy_(torch.from_numpy(beta_p))
        m.gamma.copy_(torch.from_numpy(gamma_p))
    tx = dev_nchw(x, dev, True)
    res = r.randn(*x.shape).astype(np.float32) if with_res else None
    tres = dev_nchw(res, dev, True) if with_res else None
    ty = m(tx, residual=tres)
    beta_e, gamma_e = O.gdn_reparam(beta_p, 1e-6), O.gdn_reparam(gamma_p, 0.0)
    y, nrm = O.gdn_fwd(x, beta_e, gamma_e, inverse)
    close(host(ty), y + (res if with_res else 0), RTOL, 1e-5, "y")
    dy = r.randn(*x.shape).astype(np.float32)
    ty.backward(dev_nchw(dy, dev))
    dx, dbe, dge = O.gdn_bwd(x, nrm, gamma_e, dy, inverse)
    close_norm(host(tx.grad), dx, 2e-4, "dx")
    close_norm(host(m.beta.grad), O.gdn_reparam_bwd(beta_p, dbe, 1e-6), 2e-4, "dbeta")
    close_norm(host(m.gamma.grad), O.gdn_reparam_bwd(gamma_p, dge, 0.0), 2e-4, "dgamma")
    if with_res:
        close(host(tres.grad), dy, 0, 0, "dres")


# conv -> GDN as ONE launch (the pairs at Components.py:10-15, 39-44): against the oracle, and bitwise
# against the same two ops run as separate launches
# (sizes large enough that the stand-alone conv launch does not split K, which reorders its sums)
@pytest.mark.parametrize("kind,ci,co,H,W,B,inverse", [("conv", 64, 64, 104, 100, 1, False),
                                                     ("conv", 32, 192, 60, 66, 2, False),
                                                     ("convT", 128, 128, 18, 21, 1, True),
                                                     ("convT", 64, 192, 16, 15, 2, True),
                                                     ("stem", 3, 128, 96, 90, 1, False)])
def test_conv_gdn_fused(env, kind, ci, co, H, W, B, inverse):
    nic, F_, O, dev = env
    from neural_image_compression_amd import layers as LY
    r = np.random.RandomState(ci * 7 + co)
    tr = kind == "convT"
    conv = (LY.ConvTranspose2d(ci, co, 5, stride=2, padding=2, output_padding=1) if tr
            else LY.Conv2d(ci, co, 5, stride=2, padding=2)).to(dev)
    g = LY.GDN(co, inverse=inverse).to(dev)
    beta_p, gamma_p = R.make_param("g.beta", (co,), 5), R.make_param("g.gamma", (co, co), 5)
    with torch.no_grad():
        g.beta.copy_(torch.from_numpy(beta_p))
        g.gamma.copy_(torch.from_numpy(gamma_p))
    seq = torch.nn.Sequential(conv, g)
    x = (r.rand(B, ci, H, W) if kind == "stem" else r.randn(B, ci, H, W)).astype(np.float32)
    outs = {}
    for fuse in (True, False):
        LY.FUSE_CONV_GDN = fuse
        try:
            tx = dev_nchw(x, dev, True)
            seq.zero_grad(set_to_none=True)
            y = LY.run_fused(seq, tx)
            dy = torch.from_numpy(np.random.RandomState(1).randn(*y.shape).astype(np.float32)).to(dev)
            y.backward(dy)
        finally:
            LY.FUSE_CONV_GDN = "auto"
        outs[fuse] = [y.detach(), tx.grad] + [p_.grad for p_ in seq.parameters()]
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b), "fused conv+GDN differs from the two-launch path"
    # inference: under no_grad the tensors only a backward pass reads (conv output, norm) are not written;
    # y is the same launch's y
    for fuse in (True, False):
        LY.FUSE_CONV_GDN = fuse
        try:
            with torch.no_grad():
                assert torch.equal(LY.run_fused(seq, dev_nchw(x, dev, False)), outs[True][0])
        finally:
            LY.FUSE_CONV_GDN = "auto"
    # oracle
    w, b = host(conv.weight), host(conv.bias)
    xc = O.convT2d_fwd(x, w, b, 2, 2, 1) if tr else O.conv2d_fwd(x, w, b, 2, 2)
    beta_e, gamma_e = O.gdn_reparam(beta_p, 1e-6), O.gdn_reparam(gamma_p, 0.0)
    y_ref, nrm = O.gdn_fwd(xc, beta_e, gamma_e, inverse)
    close_norm(host(outs[True][0]), y_ref, 1e-4, "y")
    dx_c, dbe, dge = O.gdn_bwd(xc, nrm, gamma_e, host(dy), inverse)
    dx, dw, db = (O.convT2d_bwd(x, w, dx_c, 2, 2, 1) if tr else O.conv2d_bwd(x, w, dx_c, 2, 2))
    close_norm(host(outs[True][1]), dx, 3e-4, "dx")
    close_norm(host(conv.weight.grad), dw, 3e-4, "dw")
    close_norm(host(conv.bias.grad), db, 3e-4, "db")
    close_norm(host(g.gamma.grad), O.gdn_reparam_bwd(gamma_p, dge, 0.0), 3e-4, "dgamma")


def test_conv_gdn_fused_rejects_unsupported(env):
    nic, F_, O, dev = env
    assert F_.fused_gdn_supported(64, 192) and F_.fused_gdn_supported(76, 128)
    assert not F_.fused_gdn_supported(64, 96) and not F_.fused_gdn_supported(64, 320)
    # "preferred" = the layers that run on 64-row tiles anyway: the RGB stem and small grids
    assert F_.fused_gdn_preferred((32, 3, 256, 256), (192, 3, 5, 5), 2, 2, False)
    assert F_.fused_gdn_preferred((32, 192, 64, 64), (192, 192, 5, 5), 2, 2, False)
    assert not F_.fused_gdn_preferred((32, 192, 128, 128), (192, 192, 5, 5), 2, 2, False)
    assert not F_.fused_gdn_preferred((32, 192, 64, 64), (192, 192, 5, 5), 2, 2, True, 1)


# ---------------------------------------------------------------------------------------------
# entropy models vs reference-pinned goldens and the oracle
# ---------------------------------------------------------------------------------------------
def _ks(fx):
    return [(k, tuple(s)) for k, s in json.loads(str(fx["keys_shapes"]))]


def _load_state(mod, fx, seed, dev):
    st = R.make_state(_ks(fx), seed)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return mod.to(dev), st


@pytest.mark.parametrize("K", [1, 3])
def test_entropy_parameters_golden(env, golden_dir, K):
    nic, F_, O, dev = env
    from neural_image_compression_amd.entropy import EntropyParameters
    fx = np.load(os.path.join(golden_dir, f"op_entropy_parameters_K{K}.npz"))
    M, B, h, w = int(fx["M"]), int(fx["B"]), int(fx["h"]), int(fx["w"])
    m, _ = _load_state(EntropyParameters(M, M, K), fx, int(fx["seed_state"]), dev)
    comb = R.make_noise((B, 4 * M, h, w), int(fx["seed_in"])) * 4 - 2
    tx = dev_nchw(comb, dev, True)
    outs = m(tx)
    loss = 0
    for i, o in enumerate(outs):
        close(host(o), fx[f"out{i}"], RTOL, 1e-6, f"out{i}")
        cot = R.make_noise(tuple(o.shape), int(fx["seed_cot"]) + i) - 0.5
        loss = loss + (o * torch.from_numpy(cot).to(dev)).sum()
    loss.backward()
    close_norm(host(tx.grad), fx["dx"], RTOL, "dx")
    for k, p in m.named_parameters():
        g = host(p.grad).ravel()
        if g.size > 8192:
            g = g[:: -(-g.size // 4096)]
        close_norm(g, fx["grad." + k], RTOL, k)


def test_factorized_golden(env, golden_dir):
    nic, F_, O, dev = env
    from neural_image_compression_amd.entropy import FactorizedEntropyBottleneck
    fx = np.load(os.path.join(golden_dir, "op_factorized.npz"))
    m, _ = _load_state(FactorizedEntropyBottleneck(int(fx["C"])), fx, int(fx["seed_state"]), dev)
    tx = dev_nchw(fx["x"], dev, True)
    p = m(tx)
    close(host(p), fx["p"], RTOL, 1e-9, "p")
    assert float(p[1, 0, 0, 0]) == np.float32(1e-9)
    close(host(m._likelihood(tx.detach())), fx["p_raw"], RTOL, 1e-12, "p_raw")
    cot = R.make_noise(tuple(p.shape), int(fx["seed_cot"])) - 0.5
    (torch.log(p) * torch.from_numpy(cot).to(dev)).sum().backward()
    close_norm(host(tx.grad), fx["dx"], RTOL, "dx")
    for k, q in m.named_parameters():
        close_norm(host(q.grad), fx["grad." + k], RTOL, k)
    xs = torch.from_numpy(fx["xs"]).to(dev)
    close(host(m.channel_cdf(2, xs)), fx["cdf_ch2"], 1e-5, 1e-7, "cdf")
    close(host(m.channel_pmf(2, xs)), fx["pmf_ch2"], RTOL, 1e-7, "pmf")


def test_factorized_fused_log_matches_oracle(env):
    nic, F_, O, dev = env
    from neural_image_compression_amd.entropy import FactorizedEntropyBottleneck
    C = 16
    m = FactorizedEntropyBottleneck(C)
    ks = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    st = R.make_state(ks, 5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    m = m.to(dev)
    x = (R.make_noise((3, C, 4, 4), 6) * 10 - 5).astype(np.float32)
    tx = dev_nchw(x, dev, True)
    p, logp = m.likelihood_and_log(tx)
    packed = O.fe_pack([st[f"matrices.{i}"] for i in range(4)], [st[f"biases.{i}"] for i in range(4)],
                       [st[f"factors.{i}"] for i in range(3)])
    op, ologp = O.factorized_fwd(x, packed)
    close(host(p), op, RTOL, 1e-9, "p")
    close(host(logp), ologp, RTOL, 1e-6, "logp")
    cot = R.make_noise(x.shape, 7) - 0.5
    (logp * torch.from_numpy(cot).to(dev)).sum().backward()
    dx, dpk = O.factorized_bwd(x, packed, None, cot)
    close_norm(host(tx.grad), dx, RTOL, "dx")
    mats, bs, fs = O.fe_unpack(dpk)
    for i in range(4):
        close_norm(host(m.matrices[i].grad), mats[i], RTOL, f"m{i}")
        close_norm(host(m.biases[i].grad), bs[i], RTOL, f"b{i}")
    for i in range(3):
        close_norm(host(m.factors[i].grad), fs[i], RTOL, f"f{i}")


def test_gaussian_golden(env, golden_dir):
    nic, F_, O, dev = env
    from neural_image_compression_amd.entropy import GaussianConditional, GaussianMixtureConditional
    fx = np.load(os.path.join(golden_dir, "op_gaussian.npz"))
    x, mu, sg = fx["x"], fx["mu"], fx["sigma"]
    tx, tm, ts = dev_nchw(x, dev, True), dev_nchw(mu, dev, True), dev_nchw(sg, dev, True)
    p1 = GaussianConditional()(tx, mu=tm, sigma=ts)
    # 1e-4 relative + the fp32 ulp floor of the CDF difference (see tests/test_oracle_golden.py)
    close(host(p1), fx["p1"], RTOL, 1.5e-7, "p1")
    cot = R.make_noise(tuple(p1.shape), int(fx["seed_cot"])) - 0.5
    (torch.log(p1) * torch.from_numpy(cot).to(dev)).sum().backward()
    sel = fx["p1"] > 2e-3
    close(host(tx.grad)[sel], fx["dx1"][sel], 5e-4, 1e-5, "dx1")
    close(host(tm.grad)[sel], fx["dmu1"][sel], 5e-4, 1e-5, "dmu1")
    close(host(ts.grad)[sel], fx["dsigma1"][sel], 5e-4, 1e-5, "dsigma1")
    dead = fx["p1"] <= 1e-9
    assert (host(tx.grad)[dead] == 0).all()
    ws, mus, sgs = fx["weights"], fx["mus"], fx["sigmas"]
    tx = dev_nchw(x, dev, True)
    tw, tm, ts = (torch.from_numpy(a).to(dev).requires_grad_(True) for a in (ws, mus, sgs))
    p3 = GaussianMixtureConditional()(tx, weights=tw, mus=tm, sigmas=ts)
    close(host(p3), fx["p3"], RTOL, 1.5e-7, "p3")
    (torch.log(p3) * torch.from_numpy(cot).to(dev)).sum().backward()
    sel = fx["p3"] > 2e-3
    close(host(tx.grad)[sel], fx["dx3"][sel], 5e-4, 1e-5, "dx3")
    sel5 = np.broadcast_to(sel[:, None], ws.shape)
    for name, t in (("dw3", tw), ("dmu3", tm), ("dsigma3", ts)):
        close(host(t.grad)[sel5], fx[name][sel5], 5e-4, 1e-5, name)


def test_rd_loss_golden(env, golden_dir):
    nic, F_, O, dev = env
    fx = np.load(os.path.join(golden_dir, "op_rd_loss.npz"))
    ty, tz, th = (dev_nchw(fx[k], dev, True) for k in ("logp_y", "logp_z", "x_hat"))
    tx = dev_nchw(fx["x"], dev)
    res = nic.rd_loss({"logp_y": ty, "logp_z": tz, "x_hat": th}, tx, float(fx["lambda_rd"]))
    for k in ("bpp_y", "bpp_z", "bpp_total", "mse", "psnr", "bits_y", "bits_z", "bits_total"):
        assert isinstance(res[k], float)
        close(res[k], float(fx[k]), 1e-5, 0, k)
    close(float(res["loss"]), float(fx["loss"]), 1e-5, 0, "loss")
    close(host(res["mse_per_image"]), fx["mse_per_image"], 1e-5, 0, "mse_img")
    close(host(res["psnr_per_image"]), fx["psnr_per_image"], 1e-5, 0, "psnr_img")
    res["loss"].backward()
    close(host(ty.grad), fx["dlogp_y"], 1e-5, 0, "dlogp_y")
    close(host(tz.grad), fx["dlogp_z"], 1e-5, 0, "dlogp_z")
    close(host(th.grad), fx["dx_hat"], RTOL, 1e-9, "dx_hat")


def test_masked_conv_golden(env, golden_dir):
    nic, F_, O, dev = env
    from neural_image_compression_amd.entropy import ContextModel
    fx = np.load(os.path.join(golden_dir, "op_masked_conv.npz"))
    m, st = _load_state(ContextModel(int(fx["M"])), fx, int(fx["seed_state"]), dev)
    tx = dev_nchw(fx["x"], dev, True)
    y = m(tx)
    close(host(y), fx["y"], RTOL, 1e-5, "y")
    assert (host(m.masked.weight) == fx["weight_after"]).all()  # in-place masking of the parameter
    cot = R.make_noise(tuple(y.shape), int(fx["seed_cot"])) - 0.5
    (y * torch.from_numpy(cot).to(dev)).sum().backward()
    close_norm(host(tx.grad), fx["dx"], RTOL, "dx")
    close_norm(host(m.masked.weight.grad), fx["dweight"], RTOL, "dweight")
    assert np.abs(host(m.masked.weight.grad) * (1 - fx["mask"])).max() > 0  # dead-tap grads unmasked
    close_norm(host(m.masked.bias.grad), fx["dbias"], RTOL, "dbias")


# ---------------------------------------------------------------------------------------------
# whole model: golden fixtures (reference modules, GDN-unpinned) and oracle at larger sizes
# ---------------------------------------------------------------------------------------------
def _sub(g):
    g = np.asarray(g, np.float32).ravel()
    if g.size > 8192:
        g = g[:: -(-g.size // 4096)]
    return g


@pytest.mark.parametrize("name", ["model_jah_M8_K1.npz", "model_jah_M8_K3.npz", "model_hmr_M8_K3.npz"])
def test_model_golden(env, golden_dir, name):
    nic, F_, O, dev = env
    fx = np.load(os.path.join(golden_dir, name))
    kind, M, K = str(fx["kind"]), int(fx["M"]), int(fx["K"])
    B, H, W, seed, lam = int(fx["B"]), int(fx["H"]), int(fx["W"]), int(fx["seed"]), float(fx["lambda_rd"])
    cls = nic.JointAutoregressiveHierarchical if kind == "5x5" else nic.HierarchicalMixtureResidual
    model, st = _load_state(cls(M, K), fx, seed, dev)
    x = R.make_image(B, H, W, seed + 1)
    tx = dev_nchw(x, dev)
    model.eval()
    with torch.no_grad():
        out = model(tx, training=False)
        res = nic.rd_loss(out, tx, lam)
    # staged parity (quantisation-flip sensitivity): latents first ...
    close(host(out["y"]), fx["eval.y"], RTOL, 1e-4, "y")
    close(host(out["z"]), fx["eval.z"], RTOL, 1e-4, "z")
    flips = int((host(out["y_in"]) != fx["eval.y_in"]).sum() + (host(out["z_in"]) != fx["eval.z_in"]).sum())
    assert flips == 0, f"{flips} rounding flips vs the reference"
    close(host(out["x_hat"]), fx["eval.x_hat"], RTOL, 1e-4, "x_hat")
    for k in ("bpp_y", "bpp_z", "bpp_total", "mse", "psnr"):
        close(res[k], float(fx["eval.loss." + k]), RTOL, 0, "eval " + k)
    for k in ("mu", "sigma", "weights", "mus", "sigmas"):
        if "eval." + k in fx.files:
            assert tuple(out[k].shape) == fx["eval." + k].shape
            close(host(out[k]), fx["eval." + k], RTOL, 1e-4, k)
    # ... then a training step with the recorded noise
    model.train()
    uz = R.make_noise(tuple(fx["train.z"].shape), seed + 2)
    uy = R.make_noise(tuple(fx["train.y"].shape), seed + 3)
    out = model(tx, noise=(dev_nchw(uz, dev), dev_nchw(uy, dev)))
    res = nic.rd_loss(out, tx, lam)
    res["loss"].backward()
    close(host(out["y_in"]), fx["train.y_in"], RTOL, 1e-4, "y_in")
    close(host(out["x_hat"]), fx["train.x_hat"], RTOL, 1e-4, "x_hat")
    close(host(out["logp_z"]), fx["train.logp_z"], RTOL, 1e-5, "logp_z")
    for k in ("bpp_y", "bpp_z", "bpp_total", "mse", "psnr"):
        close(res[k], float(fx["train.loss." + k]), RTOL, 0, "train " + k)
    close(float(res["loss"]), float(fx["train.loss.loss"]), RTOL, 0, "loss")
    checked = 0
    for pk, p in model.named_parameters():
        ref = fx["grad." + pk]
        g = _sub(host(p.grad))
        gn = float(fx["gradnorm." + pk])
        scale = max(np.abs(ref).max(), gn / np.sqrt(max(p.numel(), 1)), 1e-12)
        assert np.abs(g - ref).max() <= 2e-4 * scale + 1e-7, (pk, np.abs(g - ref).max(), scale)
        checked += 1
    assert checked > 30


@pytest.mark.parametrize("kind,M,K,B,H,W", [("5x5", 32, 1, 2, 64, 128), ("5x5", 48, 3, 1, 64, 64),
                                            ("3x3", 32, 3, 1, 64, 64),
                                            # ragged: channel counts off the 16/32-wide tiles (scalar-load
                                            # variant, zero-padded K and N), odd batch, tall image
                                            ("5x5", 20, 2, 3, 128, 64), ("5x5", 7, 1, 1, 64, 64),
                                            ("3x3", 12, 1, 2, 64, 128)])
def test_model_vs_oracle(env, kind, M, K, B, H, W):
    """Bigger channel counts (vector-load fast path, multi-tile N) and ragged shapes against the oracle."""
    nic, F_, O, dev = env
    cls = nic.JointAutoregressiveHierarchical if kind == "5x5" else nic.HierarchicalMixtureResidual
    model = cls(M, K)
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    st = R.make_state(ks, 33)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(dev)
    x = R.make_image(B, H, W, 34)
    uz, uy = R.make_noise((B, M, H // 64, W // 64), 35), R.make_noise((B, M, H // 16, W // 16), 36)
    tx = dev_nchw(x, dev)
    out = model(tx, noise=(dev_nchw(uz, dev), dev_nchw(uy, dev)))
    res = nic.rd_loss(out, tx, 0.01)
    res["loss"].backward()
    o_out, o_loss, o_grads = O.model_forward(dict(st), x, M, K, kind, training=True, noise=(uz, uy),
                                             lambda_rd=0.01, backward=True)
    close(host(out["y"]), o_out["y"], RTOL, 1e-4, "y")
    close(host(out["z"]), o_out["z"], RTOL, 1e-4, "z")
    close(host(out["x_hat"]), o_out["x_hat"], RTOL, 1e-4, "x_hat")
    for k in ("bpp_y", "bpp_z", "bpp_total", "mse", "psnr"):
        close(res[k], o_loss[k], RTOL, 0, k)
    for pk, p in model.named_parameters():
        ref = o_grads[pk]
        scale = max(np.abs(ref).max(), np.sqrt((ref.astype(np.float64) ** 2).mean()), 1e-12)
        err = np.abs(host(p.grad) - ref).max()
        assert err <= 3e-4 * scale + 1e-7, (pk, err, scale)


def _pack_layout_ref(w, taps, K, N, s_tap, s_k, s_n):
    """numpy restatement of include/lic.h's fp32 packed layout [tap][K/16][Npad/32][2][64][4]"""
    flat = w.reshape(-1)
    npad = 32 if N <= 32 else (N + 63) // 64 * 64     # whole 64-column wave pairs (lic_common.h lic_npad_f32)
    cpt, ntile = (K + 15) // 16, npad // 32
    out = np.zeros((taps, cpt, ntile, 2, 64, 4), np.float32)
    for tap in range(taps):
        for k in range(K):
            for n in range(N):
                cb, kl = divmod(k, 16)
                tile, nl = divmod(n, 32)
                lh, r = divmod(kl, 8)
                q, e = divmod(r, 4)
                out[tap, cb, tile, q, lh * 32 + nl, e] = flat[tap * s_tap + k * s_k + n * s_n]
    return out.reshape(-1)


@pytest.mark.parametrize("d0,d1,kh", [(80, 75, 3), (64, 96, 5), (33, 20, 1), (192, 192, 5), (48, 48, 5), (16, 40, 3)])
def test_pack_weight_tiled_equals_generic(env, d0, d1, kh, monkeypatch):
    """the LDS-tiled packer (conv-weight strides) writes the same bytes as the generic gather kernel,
    and both follow the documented layout (ragged K / N zero padded)"""
    nic, F_, O, dev = env
    rng = np.random.default_rng(5)
    w = rng.standard_normal((d0, d1, kh, kh)).astype(np.float32)
    wd = torch.from_numpy(w).to(dev)
    taps = kh * kh
    for dgrad in (False, True):
        for transposed in (False, True):
            monkeypatch.delenv("LIC_PACK_NO_TILED", raising=False)
            a = host(F_._pack_conv_weight(wd, transposed, dgrad))
            monkeypatch.setenv("LIC_PACK_NO_TILED", "1")
            b = host(F_._pack_conv_weight(wd, transposed, dgrad))
            assert np.array_equal(a, b)
            if d0 * d1 * taps <= 80 * 75 * 9:
                s_a, s_b = d1 * taps, taps           # strides of dims 0 / 1
                cin_s, cout_s = (s_a, s_b) if transposed else (s_b, s_a)
                Kc, Nc = ((d0, d1) if transposed else (d1, d0))   # forward: K = Cin, N = Cout
                if dgrad:
                    ref = _pack_layout_ref(w, taps, Nc, Kc, 1, cout_s, cin_s)
                else:
                    ref = _pack_layout_ref(w, taps, Kc, Nc, 1, cin_s, cout_s)
                assert np.array_equal(a, ref)
    monkeypatch.delenv("LIC_PACK_NO_TILED", raising=False)
    m = torch.from_numpy(rng.standard_normal((d0, d1)).astype(np.float32)).to(dev)
    a = host(F_._pack_dense(m))
    monkeypatch.setenv("LIC_PACK_NO_TILED", "1")
    assert np.array_equal(a, host(F_._pack_dense(m)))


def test_missing_cuda_input_raises(env):
    nic, F_, O, dev = env
    from neural_image_compression_amd._lib import LicError
    with pytest.raises(LicError):
        F_.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 3, 3), None, 1, 1)
