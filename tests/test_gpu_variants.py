"""Every MFMA kernel VARIANT the full-size workloads dispatch, in front of the oracle.

The tile lic_igemm / lic_wgrad pick depends on the batch (a 128-row tile needs >= 512 workgroups), so
small parity shapes never reach the variants config 2 spends its time in.  Three layers of tests close
that gap ("what's weak" 1 + 2):

 1. forced tiles (descriptor fields force_bm / force_tn / force_split, force_tm / ...) on small ragged
    shapes against the plain-C oracle `O.conv2d_* / O.convT2d_*`: P % 128 != 0, chunk counts with every
    residue mod 3 (the LDS-DMA ring is unrolled by three), 4-phase transposed launches with phase
    sorting (`pgroup`), split-K slices that begin mid-tap, ragged channel counts;
 2. the real layers and the real configurations at FULL size (B = 32, 256x256; 16 x 512x512) against
    `oracle/torch_ref.py` (torch CPU ops: the arithmetic library of the reference's own CPU path);
 3. `test_every_dispatched_variant_was_checked`: the set of kernel names the cfg 2 / 3 / 3k / 5 training
    steps dispatch (collected live through lic_*_kernel_name) must be a subset of the names the
    oracle-checked tests of this file ran.

fp32 tolerance: 1e-4 relative (north star); bf16-storage tolerances are declared where used.
Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import math
import os

import numpy as np
import pytest
import torch

import golden_recipe as R

pytestmark = pytest.mark.gpu

RTOL = 1e-4
CHECKED = set()   # kernel names that ran inside an oracle comparison of this module


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    from neural_image_compression_amd import functional as F_
    from neural_image_compression_amd import _lib
    from oracle import oracle as O
    _lib.load()
    return nic, F_, O, torch.device("cuda:0")


class traced:
    """context: record the kernel variants launched inside (they are being compared with the oracle)"""

    def __init__(self, F_, igemm=None, wgrad=None):
        self.F_, self.igemm, self.wgrad = F_, igemm, wgrad

    def __enter__(self):
        self.F_.KERNEL_TRACE = set()
        self.F_.FORCE_IGEMM, self.F_.FORCE_WGRAD = self.igemm, self.wgrad
        return self

    def __exit__(self, *exc):
        self.names = self.F_.KERNEL_TRACE
        if exc[0] is None:
            CHECKED.update(self.names)
        self.F_.KERNEL_TRACE = None
        self.F_.FORCE_IGEMM = self.F_.FORCE_WGRAD = None
        return False


def dev_nchw(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if t.dim() == 4:
        t = t.contiguous(memory_format=torch.channels_last)
    if grad:
        t.requires_grad_(True)
    return t


def host(t):
    return t.detach().float().cpu().contiguous().numpy()


def close(a, b, rtol=RTOL, atol=1e-5, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), f"{what}: {bad.sum()} bad, max err {err.max():.3e} (|ref| max {np.abs(b).max():.3e})"


def close_norm(a, b, rtol=RTOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= rtol * scale, f"{what}: {np.abs(a - b).max():.3e} vs scale {scale:.3e}"


def tap_mask_a(k):
    live = 0
    for r in range(k):
        for s in range(k):
            if r < k // 2 or (r == k // 2 and s < k // 2):
                live |= 1 << (r * k + s)
    return live


# ---------------------------------------------------------------------------------------------
# 1a. lic_igemm: every (BM, TN) tile, forward + data gradient, small ragged shapes, C oracle
# ---------------------------------------------------------------------------------------------
TILES = [(64, 1), (64, 2), (64, 3), (128, 1), (128, 2), (128, 3)]
# kind, k, stride, pad, out_pad, H, W, B: P % 128 != 0 everywhere
IGEMM_GEOMS = [
    ("conv", 5, 2, 2, 0, 18, 10, 3),    # 3 * 9 * 5 = 135 output pixels
    ("convT", 5, 2, 2, 1, 7, 5, 3),     # four phases of 105 pixels, 9/6/6/4 taps
    ("conv", 3, 1, 1, 0, 9, 11, 2),     # 198 pixels
    ("masked", 5, 1, 2, 0, 12, 7, 2),   # 12 live taps of 25 (tap_mask)
    ("conv", 1, 1, 0, 0, 13, 11, 1),    # 1x1: 4 / 8 / 12 chunks
]


@pytest.mark.parametrize("bm,tn", TILES)
@pytest.mark.parametrize("kind,k,s,p,op,H,W,B", IGEMM_GEOMS)
def test_igemm_forced_tile_fwd_bwd(env, bm, tn, kind, k, s, p, op, H, W, B):
    """Cin = Cout = 64*tn (so the data gradient runs the same forced tile): chunk counts per phase are
    taps * 4 * tn -- 100, 200, 300 for the 5x5 layers (residues 1, 2, 0 of the ring unrolled by 3)."""
    nic, F_, O, dev = env
    C = 64 * tn
    r = np.random.RandomState(1000 * bm + 10 * tn + k + H)
    x = r.randn(B, C, H, W).astype(np.float32)
    b = r.randn(C).astype(np.float32)
    w = (r.randn(C, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    tx, tb = dev_nchw(x, dev, True), dev_nchw(b, dev, True)
    tw = dev_nchw(w, dev).contiguous().requires_grad_(True)
    with traced(F_, igemm=(bm, tn, 0)) as tr:
        if kind == "convT":
            ty = F_.conv_transpose2d(tx, tw, tb, s, p, op)
            y = O.convT2d_fwd(x, w, b, s, p, op)
        elif kind == "masked":
            wm = w * O.mask_a(w.shape)
            with torch.no_grad():
                tw.mul_(dev_nchw(O.mask_a(w.shape), dev).contiguous())
            ty = F_.conv2d(tx, tw, tb, s, p, False, 0.01, tap_mask_a(k))
            y = O.conv2d_fwd(x, wm, b, s, p)
        else:
            ty = F_.conv2d(tx, tw, tb, s, p)
            y = O.conv2d_fwd(x, w, b, s, p)
        close(host(ty), y, RTOL, 1e-5, "y")
        dy = r.randn(*y.shape).astype(np.float32)
        ty.backward(dev_nchw(dy, dev))
        if kind == "convT":
            dx, dw, db = O.convT2d_bwd(x, w, dy, s, p, op)
        else:
            dx, dw, db = O.conv2d_bwd(x, wm if kind == "masked" else w, dy, s, p)
        close_norm(host(tx.grad), dx, RTOL, "dx")
        close_norm(host(tw.grad), dw, RTOL, "dw")   # the masked conv's weight gradient is NOT masked
        close_norm(host(tb.grad), db, RTOL, "db")
    full = "true"
    assert f"igemm_kernel<{bm}, {tn}, true, {full}, false, true>" in tr.names, tr.names


@pytest.mark.parametrize("bm,tn", TILES)
@pytest.mark.parametrize("cin,cout_mult", [(16, 1.0), (20, 1.0), (48, 2.0), (36, 0.9375)])
def test_igemm_forced_tile_ragged_k_and_n(env, bm, tn, cin, cout_mult):
    """Forward only (the data gradient would need Cin to tile as well): Cin of 1 / 1.25 / 3 / 2.25 chunks per
    tap (partial last chunk: the DMA's `ci < Cin` zero page), Cout of two N tiles, and Cout = 60*tn whose
    32-padding still fills the tile (stores guarded by col < Cout)."""
    nic, F_, O, dev = env
    cout = int(64 * tn * cout_mult)
    r = np.random.RandomState(cin * 7 + cout)
    x = r.randn(2, cin, 11, 9).astype(np.float32)
    w = (r.randn(cout, cin, 5, 5) / np.sqrt(cin * 25)).astype(np.float32)
    b = r.randn(cout).astype(np.float32)
    with traced(F_, igemm=(bm, tn, 0)), torch.no_grad():
        ty = F_.conv2d(dev_nchw(x, dev), dev_nchw(w, dev).contiguous(), dev_nchw(b, dev), 2, 2, True)
        close(host(ty), O.leaky_relu_fwd(O.conv2d_fwd(x, w, b, 2, 2)), RTOL, 1e-5, "conv y")
        wt = np.ascontiguousarray(w.transpose(1, 0, 2, 3))[:cin]  # [Cin, Cout, 5, 5] for the transposed layer
        ty = F_.conv_transpose2d(dev_nchw(x, dev), dev_nchw(wt, dev).contiguous(), dev_nchw(b, dev), 2, 2, 1)
        close(host(ty), O.convT2d_fwd(x, wt, b, 2, 2, 1), RTOL, 1e-5, "convT y")


@pytest.mark.parametrize("bm,tn", [(128, 3), (128, 1), (64, 2)])
def test_igemm_phase_sorted_transposed_launch(env, bm, tn):
    """>= 128 M tiles per phase switches the 4-phase launch to phase-sorted groups of 64 tiles (`pgroup`,
    MT padded to whole groups, the padding tiles exit at once): 2 x 96 x 96 input pixels per phase."""
    nic, F_, O, dev = env
    from oracle import torch_ref as TR
    cin, cout, B, Hh = 16, 64 * tn, 2, 96
    if bm == 64:
        Hh = 72  # 2*72*72 / 64 = 162 tiles
    r = np.random.RandomState(bm + tn)
    x = r.randn(B, cin, Hh, Hh).astype(np.float32)
    w = (r.randn(cin, cout, 5, 5) / np.sqrt(cin * 25 / 4)).astype(np.float32)
    b = r.randn(cout).astype(np.float32)
    with traced(F_, igemm=(bm, tn, 0)), torch.no_grad():
        ty = F_.conv_transpose2d(dev_nchw(x, dev), dev_nchw(w, dev).contiguous(), dev_nchw(b, dev), 2, 2, 1)
        ref = TR.conv_transpose2d(x, w, b, 2, 2, 1)
    close(host(ty), ref, RTOL, 1e-5, "convT y (phase-sorted)")


@pytest.mark.parametrize("bm,tn,split", [(64, 1, 4), (128, 1, 5), (64, 3, 2), (128, 3, 7), (64, 2, 3)])
def test_igemm_split_k_mid_tap(env, bm, tn, split):
    """K splits that begin in the middle of a tap (3x3 taps x 3 chunks = 27 chunks over `split` slices) and
    the fixed-order reduction behind them, with the fused LeakyReLU of the hyper layers."""
    nic, F_, O, dev = env
    cin, cout = 48, 64 * tn
    r = np.random.RandomState(split)
    x = r.randn(2, cin, 9, 7).astype(np.float32)
    w = (r.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float32)
    b = r.randn(cout).astype(np.float32)
    with traced(F_, igemm=(bm, tn, split)), torch.no_grad():
        ty = F_.conv2d(dev_nchw(x, dev), dev_nchw(w, dev).contiguous(), dev_nchw(b, dev), 1, 1, True)
        close(host(ty), O.leaky_relu_fwd(O.conv2d_fwd(x, w, b, 1, 1)), RTOL, 1e-5, "y")
        wt = np.ascontiguousarray(w.transpose(1, 0, 2, 3))
        ty = F_.conv_transpose2d(dev_nchw(x, dev), dev_nchw(wt, dev).contiguous(), dev_nchw(b, dev), 2, 1, 1, True)
        close(host(ty), O.leaky_relu_fwd(O.convT2d_fwd(x, wt, b, 2, 1, 1)), RTOL, 1e-5, "convT y")


# ---------------------------------------------------------------------------------------------
# 1b. lic_wgrad: every tile shape (incl. ragged channel counts, squared operand), C oracle
# ---------------------------------------------------------------------------------------------
WGRAD_CASES = [
    # (tm, tn), Cin (rows), Cout (cols)
    ((1, 1), 24, 40), ((1, 1), 64, 64), ((1, 3), 64, 192), ((1, 3), 48, 180), ((2, 1), 128, 64),
    ((2, 2), 128, 128), ((2, 2), 100, 120), ((2, 3), 128, 192), ((2, 3), 288, 384), ((2, 3), 288, 180),
    ((3, 3), 192, 192), ((3, 3), 384, 192),
]


@pytest.mark.parametrize("tile,cin,cout", WGRAD_CASES)
@pytest.mark.parametrize("transposed", [False, True])
def test_wgrad_forced_tile(env, tile, cin, cout, transposed):
    """5x5 stride-2 layers: Cm = Cin, Cn = Cout for conv and convT alike; 3 splits of the pixel range."""
    nic, F_, O, dev = env
    r = np.random.RandomState(cin + cout)
    B, H, W = 2, 10, 6
    x = r.randn(B, cin, H, W).astype(np.float32)
    with traced(F_, wgrad=(tile[0], tile[1], 3)) as tr:
        if transposed:
            w = (r.randn(cin, cout, 5, 5) / np.sqrt(cin * 25)).astype(np.float32)
            tx, tw = dev_nchw(x, dev, True), dev_nchw(w, dev).contiguous().requires_grad_(True)
            ty = F_.conv_transpose2d(tx, tw, None, 2, 2, 1)
            dy = r.randn(*ty.shape).astype(np.float32)
            ty.backward(dev_nchw(dy, dev))
            _, dw, _ = O.convT2d_bwd(x, w, dy, 2, 2, 1)
        else:
            w = (r.randn(cout, cin, 5, 5) / np.sqrt(cin * 25)).astype(np.float32)
            tx, tw = dev_nchw(x, dev, True), dev_nchw(w, dev).contiguous().requires_grad_(True)
            ty = F_.conv2d(tx, tw, None, 2, 2)
            dy = r.randn(*ty.shape).astype(np.float32)
            ty.backward(dev_nchw(dy, dev))
            _, dw, _ = O.conv2d_bwd(x, w, dy, 2, 2)
        close_norm(host(tw.grad), dw, RTOL, "dw")
    assert any(n.startswith(f"wgrad_glds_kernel<{tile[0]}, {tile[1]},") for n in tr.names), tr.names


@pytest.mark.parametrize("C,tile", [(64, (1, 1)), (128, (2, 2)), (192, (1, 3)), (192, (2, 3)), (192, (3, 3))])
@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_dgamma_forced_tile(env, C, tile, inverse):
    """GDN / IGDN parameter gradients: d gamma = t^T . x^2 runs lic_wgrad with the squared column operand
    (`wgrad_glds_kernel<.., true, ..>`)."""
    nic, F_, O, dev = env
    from neural_image_compression_amd.layers import GDN
    r = np.random.RandomState(C + int(inverse))
    g = GDN(C, inverse=inverse).to(dev)
    with torch.no_grad():
        g.gamma.add_(torch.from_numpy((0.02 * r.rand(C, C)).astype(np.float32)).to(dev))
    x = r.randn(2, C, 9, 7).astype(np.float32)
    dy = r.randn(2, C, 9, 7).astype(np.float32)
    tx = dev_nchw(x, dev, True)
    with traced(F_, wgrad=(tile[0], tile[1], 2)) as tr:
        ty = g(tx)
        ty.backward(dev_nchw(dy, dev))
        beta_p, gamma_p = host(g.beta), host(g.gamma)
        beta_e, gamma_e = O.gdn_reparam(beta_p, 1e-6), O.gdn_reparam(gamma_p, 0.0)
        y, norm = O.gdn_fwd(x, beta_e, gamma_e, inverse)
        dx, dbe, dge = O.gdn_bwd(x, norm, gamma_e, dy, inverse)
        close(host(ty), y, RTOL, 1e-5, "y")
        close_norm(host(tx.grad), dx, RTOL, "dx")
        close_norm(host(g.gamma.grad), O.gdn_reparam_bwd(gamma_p, dge, 0.0), RTOL, "dgamma")
        close_norm(host(g.beta.grad), O.gdn_reparam_bwd(beta_p, dbe, 1e-6), RTOL, "dbeta")
    assert any(n.startswith(f"wgrad_glds_kernel<{tile[0]}, {tile[1]}, true") for n in tr.names), tr.names


# ---------------------------------------------------------------------------------------------
# 2a. the real layers at the size config 2 runs them, against torch CPU ops
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H", [(16, 128), (32, 64), (32, 32)])
def test_real_5x5_layer_pair_vs_torch_cpu(env, B, H):
    """Components.py:12,41: Conv2d(192,192,5,2,2) at H^2 -> (H/2)^2 and its transposed twin
    ConvTranspose2d(192,192,5,2,2,1) back, forward + all three gradients, automatic tile choice
    (B = 16 at 128^2 is the smallest batch that still takes the 128x192 LDS-DMA tile and the 192x192 wgrad)."""
    nic, F_, O, dev = env
    from oracle import torch_ref as TR
    r = np.random.RandomState(H)
    C = 192
    x = r.randn(B, C, H, H).astype(np.float32)
    w = (r.randn(C, C, 5, 5) / np.sqrt(C * 25)).astype(np.float32)
    b = r.randn(C).astype(np.float32)
    dy = r.randn(B, C, H // 2, H // 2).astype(np.float32)
    with traced(F_) as tr:
        tx, tb = dev_nchw(x, dev, True), dev_nchw(b, dev, True)
        tw = dev_nchw(w, dev).contiguous().requires_grad_(True)
        ty = F_.conv2d(tx, tw, tb, 2, 2)
        ty.backward(dev_nchw(dy, dev))
        y, dx, dw, db = TR.conv2d_step(x, w, b, dy, 2, 2)
        close(host(ty), y, RTOL, 1e-4, "conv y")
        close_norm(host(tx.grad), dx, RTOL, "conv dx")
        close_norm(host(tw.grad), dw, 2e-4, "conv dw")   # sums over up to 65536 pixels
        close_norm(host(tb.grad), db, 2e-4, "conv db")
        del tx, ty
        # transposed twin: dy-shaped input back up to H x H
        wt = (r.randn(C, C, 5, 5) / np.sqrt(C * 25 / 4)).astype(np.float32)
        tx2, tb2 = dev_nchw(dy, dev, True), dev_nchw(b, dev, True)
        tw2 = dev_nchw(wt, dev).contiguous().requires_grad_(True)
        ty2 = F_.conv_transpose2d(tx2, tw2, tb2, 2, 2, 1)
        ty2.backward(dev_nchw(x, dev))
        y2, dx2, dw2, db2 = TR.conv_transpose2d_step(dy, wt, b, x, 2, 2, 1)
        close(host(ty2), y2, RTOL, 1e-4, "convT y")
        close_norm(host(tx2.grad), dx2, RTOL, "convT dx")
        close_norm(host(tw2.grad), dw2, 2e-4, "convT dw")
        close_norm(host(tb2.grad), db2, 2e-4, "convT db")
    if (B, H) == (16, 128):
        assert "igemm_kernel<128, 3, true, true, false, true>" in tr.names, tr.names
        assert "wgrad_glds_kernel<3, 3, false, true>" in tr.names, tr.names


# ---------------------------------------------------------------------------------------------
# 2b. whole configurations at FULL size against oracle/torch_ref.py
# ---------------------------------------------------------------------------------------------
def _full_step(nic, dev, M, K, B, H, W, seed, precision="fp32", lam=0.01, cls=None):
    model = (cls or nic.JointAutoregressiveHierarchical)(M, K)
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    st = R.make_state(ks, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(dev)
    if precision != "fp32":
        model.set_precision(precision)
    x = R.make_image(B, H, W, seed + 1)
    uz, uy = R.make_noise((B, M, H // 64, W // 64), seed + 2), R.make_noise((B, M, H // 16, W // 16), seed + 3)
    tx = torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last)
    out = model(tx, noise=(torch.from_numpy(uz).to(dev), torch.from_numpy(uy).to(dev)))
    res = nic.rd_loss(out, tx, lam)
    res["loss"].backward()
    torch.cuda.synchronize()
    return model, st, x, (uz, uy), out, res


def _compare_fp32(model, out, res, t_out, t_loss, t_grads, grad_tol=5e-4):
    for k in ("y", "z", "x_hat"):
        a, b = host(out[k]).astype(np.float64), t_out[k].astype(np.float64)
        assert (np.abs(a - b) <= 1e-4 + 1e-4 * np.abs(b)).all(), (k, np.abs(a - b).max())
    for k in ("bpp_y", "bpp_z", "bpp_total", "mse", "psnr"):
        assert abs(res[k] - t_loss[k]) <= 1e-4 * abs(t_loss[k]), (k, res[k], t_loss[k])
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        ref = t_grads[name]
        err = float(np.abs(host(p.grad) - ref).max())
        e = max(0.0, err - 3e-7) / max(np.abs(ref).max(), 1e-12)  # (3e-7 floor: see test_gpu_fullsize.py)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] <= grad_tol, worst


CFG_FP32 = {
    # name: (M, K, B, H, W)  -- BASELINE.json configs; "5" = the serial-context stress shape
    "cfg2": (192, 1, 32, 256, 256),
    "cfg3k": (128, 3, 32, 256, 256),
    "cfg4": (192, 3, 32, 256, 256),
    "cfg5": (192, 3, 16, 512, 512),
}


@pytest.mark.parametrize("name", list(CFG_FP32))
def test_full_size_config_step_vs_torch_cpu_path(env, name):
    """The exact workload of a BASELINE config -- batch, image size, capacity, K -- forward + rd_loss +
    backward on the HIP path against the torch-CPU restatement of the reference: latents, x_hat, bpp,
    PSNR within 1e-4 relative and every parameter gradient relative to its tensor's scale.  Whatever tile
    variants the bench dispatches for this config are the ones compared here."""
    nic, F_, O, dev = env
    from oracle import torch_ref as TR
    M, K, B, H, W = CFG_FP32[name]
    with traced(F_):
        model, st, x, noise, out, res = _full_step(nic, dev, M, K, B, H, W, 300 + M + K)
        t_out, t_loss, t_grads = TR.step(st, x, M, K, "5x5", noise, 0.01)
        _compare_fp32(model, out, res, t_out, t_loss, t_grads)


def test_hmr_full_capacity_step_vs_torch_cpu_path(env):
    """SURVEY 8(a) row a5 at FULL capacity: HierarchicalMixtureResidual(192, K=3) -- the 3x3 residual stacks
    (Components.py:20-32,49-62,77-91,107-122; Layers.py:18-119) -- 4x3x256x256, forward + rd_loss + backward
    against the torch-CPU restatement, same tolerances as the 5x5 model."""
    nic, F_, O, dev = env
    from oracle import torch_ref as TR
    M, K, B = 192, 3, 4
    with traced(F_):
        model, st, x, noise, out, res = _full_step(nic, dev, M, K, B, 256, 256, 610, cls=nic.HierarchicalMixtureResidual)
        t_out, t_loss, t_grads = TR.step(st, x, M, K, "3x3", noise, 0.01)
        _compare_fp32(model, out, res, t_out, t_loss, t_grads)


def test_cfg5_one_image_train_step_vs_c_oracle(env):
    """Config 5's geometry (512x512: y 32x32, z 8x8, masked conv at 32^2, K = 3) for ONE image against the
    plain-C oracle (the second, independent restatement)."""
    nic, F_, O, dev = env
    M, K = 192, 3
    with traced(F_):
        model, st, x, noise, out, res = _full_step(nic, dev, M, K, 1, 512, 512, 555)
        o_out, o_loss, o_grads = O.model_forward(dict(st), x, M, K, "5x5", training=True, noise=noise,
                                                 lambda_rd=0.01, backward=True)
        # Gradient tolerance 2e-3 of each tensor's scale for this ONE-image case: the gradients of
        # entropy_parameters.net.2 are ~3e-3 sums of cancelling terms here, and an fp64 evaluation of the same
        # step (torch CPU, float64) sits 8.7e-4 of that scale away from BOTH fp32 CPU restatements (C oracle
        # and torch fp32, which differ from each other by 6.7e-4): fp32 evaluation noise, not a kernel error.
        # The 16-image config above, where the sums are longer and the cancellation milder, holds 5e-4.
        _compare_fp32(model, out, res, o_out, o_loss, o_grads, grad_tol=2e-3)


# bands of the bf16-storage mode at full size: 2x the deviations measured on an MI355X (profiles/r03_bf16_deviations.json)
BF16_BANDS = {
    # measured (cfg 3): bpp_y 1.3e-4, bpp_z 9e-8, bpp_total 1.0e-4, mse 2.8e-5, PSNR 1.2e-4 dB, y 6.4e-3, cosine 0.99998
    128: {"bpp_y": 3e-4, "bpp_z": 1e-5, "bpp_total": 2.5e-4, "mse": 1e-4, "psnr_db": 5e-4, "y_rel_max": 0.013, "cosine": 0.9999},
    # measured (cfg 2h): bpp_y 1.3e-4, bpp_z 1.2e-7, bpp_total 1.1e-4, mse 3.4e-4, PSNR 1.5e-3 dB, y 7.2e-3, cosine 0.99997
    192: {"bpp_y": 3e-4, "bpp_z": 1e-5, "bpp_total": 2.5e-4, "mse": 7e-4, "psnr_db": 3e-3, "y_rel_max": 0.015, "cosine": 0.9999},
}


@pytest.mark.parametrize("M,K", [(128, 3), (192, 1)])
def test_full_size_bf16_storage_step_vs_torch_cpu_path(env, M, K):
    """Config 3 (JAH(128, K=3), B = 32, 256x256, bf16 storage in the conv/GDN stacks) and config 2's model in
    the same mode, at full size, against the fp32 torch-CPU path.  The mode is not in the reference, so
    its tolerances are OURS -- but measured, not declared: the achieved deviations are recorded
    (profiles/r03_bf16_deviations.json) and the bands are 2x those (BF16_BANDS above): bpp within 3e-4, mse within
    1e-4 / 7e-4, PSNR within 5e-4 / 3e-3 dB, every parameter gradient with more than 1e-3 of the largest gradient's
    norm has cosine >= 0.9999 with the fp32 gradient."""
    nic, F_, O, dev = env
    from oracle import torch_ref as TR
    with traced(F_):
        model, st, x, noise, out, res = _full_step(nic, dev, M, K, 32, 256, 256, 700 + M, precision="bf16")
        t_out, t_loss, t_grads = TR.step(st, x, M, K, "5x5", noise, 0.01)
        dev_rel = {k: abs(res[k] - t_loss[k]) / abs(t_loss[k]) for k in ("bpp_y", "bpp_z", "bpp_total", "mse")}
        dpsnr = abs(res["psnr"] - t_loss["psnr"])
        ymax = np.abs(t_out["y"]).max()
        dy = float(np.abs(host(out["y"]) - t_out["y"]).max() / ymax)
        norms = {n: float(np.linalg.norm(t_grads[n])) for n, _ in model.named_parameters()}
        big = max(norms.values())
        worst = ("", 1.0)
        for n, p in model.named_parameters():
            if norms[n] < 1e-3 * big:
                continue
            a, b = host(p.grad).ravel().astype(np.float64), t_grads[n].ravel().astype(np.float64)
            cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
            if cos < worst[1]:
                worst = (n, cos)
        # the achieved deviations are recorded (gpurun_out/bf16_deviations_M*.json -> DESIGN.md); the bands below are
        # 2x what was measured on an MI355X at this size, not free-standing declarations
        rec = {"M": M, "K": K, "rel": dev_rel, "psnr_db": dpsnr, "y_rel_max": dy, "worst_grad_cosine": worst}
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out_dir):
            import json
            with open(os.path.join(out_dir, f"bf16_deviations_M{M}.json"), "w") as f:
                json.dump(rec, f, indent=1)
        band = BF16_BANDS[M]
        for k, v in dev_rel.items():
            assert v <= band[k], (k, v, band[k], res[k], t_loss[k])
        assert dpsnr <= band["psnr_db"], (dpsnr, res["psnr"], t_loss["psnr"])
        assert dy <= band["y_rel_max"], dy
        assert worst[1] >= band["cosine"], worst


def test_hmr_bf16_storage_step_vs_torch_cpu_path(env):
    """The 3x3 residual model (Layers.py:27-119) in bf16 storage at full capacity --
    HierarchicalMixtureResidual(192, K=3), 4x3x256x256 -- against the fp32 torch-CPU path: rates, distortion and the
    direction of every sizeable gradient.  The bands are this mode's own (it is not in the reference): the 3x3 model has
    ~3x the layers of the 5x5 one between image and latent, each rounding its output to bf16, so the measured
    deviations (recorded in gpurun_out/bf16_deviations_hmr.json) are larger than config 3's; asserted at ~2x measured."""
    nic, F_, O, dev = env
    from oracle import torch_ref as TR
    M, K, B = 192, 3, 4
    with traced(F_):
        model, st, x, noise, out, res = _full_step(nic, dev, M, K, B, 256, 256, 611, cls=nic.HierarchicalMixtureResidual,
                                                   precision="bf16")
        t_out, t_loss, t_grads = TR.step(st, x, M, K, "3x3", noise, 0.01)
        dev_rel = {k: abs(res[k] - t_loss[k]) / abs(t_loss[k]) for k in ("bpp_y", "bpp_z", "bpp_total", "mse")}
        dpsnr = abs(res["psnr"] - t_loss["psnr"])
        dy = float(np.abs(host(out["y"]) - t_out["y"]).max() / np.abs(t_out["y"]).max())
        norms = {n: float(np.linalg.norm(t_grads[n])) for n, _ in model.named_parameters()}
        big = max(norms.values())
        worst = ("", 1.0)
        for n, p in model.named_parameters():
            if norms[n] < 1e-3 * big:
                continue
            a, b = host(p.grad).ravel().astype(np.float64), t_grads[n].ravel().astype(np.float64)
            cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
            if cos < worst[1]:
                worst = (n, cos)
        rec = {"model": "hmr", "M": M, "K": K, "rel": dev_rel, "psnr_db": dpsnr, "y_rel_max": dy, "worst_grad_cosine": worst}
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out_dir):
            import json
            with open(os.path.join(out_dir, "bf16_deviations_hmr.json"), "w") as f:
                json.dump(rec, f, indent=1)
        band = HMR_BF16_BAND
        for k, v in dev_rel.items():
            assert v <= band[k], (k, v, band[k], res[k], t_loss[k])
        assert dpsnr <= band["psnr_db"], (dpsnr, res["psnr"], t_loss["psnr"])
        assert dy <= band["y_rel_max"], dy
        assert worst[1] >= band["cosine"], worst


# ~2x the deviations measured on an MI355X (profiles/r03_bf16_deviations.json, "hmr": bpp_y 2.3e-4, bpp_z 0, bpp_total
# 1.9e-4, mse 1.2e-3, PSNR 5.1e-3 dB, y 8.3e-3, worst gradient cosine 0.99995 at encoder.net.3.conv1.weight)
HMR_BF16_BAND = {"bpp_y": 5e-4, "bpp_z": 1e-5, "bpp_total": 4e-4, "mse": 2.5e-3, "psnr_db": 1e-2, "y_rel_max": 0.017,
                 "cosine": 0.9999}


# ---------------------------------------------------------------------------------------------
# 3. nothing the benchmark dispatches is left unchecked
# ---------------------------------------------------------------------------------------------
def test_every_dispatched_variant_was_checked(env):
    """Collect, through lic_*_kernel_name, the MFMA kernel variants one training step of bench.py's cfg 2,
    3, 3k, 4 and 5 dispatches with default-initialised weights and the bench's own input recipe, and
    require each of them to have run inside an oracle comparison above (same process, same library)."""
    nic, F_, O, dev = env
    import bench
    assert CHECKED, "run the whole module: this test audits the tests above"
    missing = {}
    for cfg in ("2", "3", "3k", "4", "5"):
        kind, M, K, B, H, W, lam = bench.CONFIGS[cfg]
        torch.manual_seed(0)
        model = nic.JointAutoregressiveHierarchical(M, K).to(dev)
        if cfg in bench.BF16_CONFIGS:
            model.set_precision("bf16")
        g = torch.Generator(device="cpu").manual_seed(1234)
        x = torch.rand(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        F_.KERNEL_TRACE = set()
        try:
            out = model(x)
            nic.rd_loss(out, x, lam, sync=False)["loss"].backward()
            torch.cuda.synchronize()
            names = F_.KERNEL_TRACE
        finally:
            F_.KERNEL_TRACE = None
        assert any("128, 3" in n for n in names) or M != 192, names
        left = sorted(n for n in names if n not in CHECKED)
        if left:
            missing[cfg] = left
        del model, out
    assert not missing, f"dispatched by the bench but never compared with the oracle: {missing}"
