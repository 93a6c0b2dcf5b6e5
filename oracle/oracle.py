"""numpy front-end of the CPU oracle (oracle/lic_oracle.c) + whole-model restatement.

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from neural_image_compression_amd/.

Model-level wiring restates (paths under /root/reference):
  Models.py:49-106 / 148-205   forward orchestration (noise z-then-y, round, cat([phi, psi]))
  Components.py:6-122          Encoder/Decoder/HyperEncoder/HyperDecoder 5x5 and 3x3 stacks
  Layers.py:18-119             TransposedDeconv3x3, ResidualBlock{,WithStride,Upsample}
  ContextModels.py:9-36        type-A masked 5x5 convolution
  ParametersModels.py:20-64    1x1-conv MLP and its activations
  EntropyModels.py:29-233      factorised / Gaussian / mixture likelihoods
  RateDistortionLoss.py:5-49   rd_loss
GDN is third-party compressai (absent offline) -> "parity unpinned", SURVEY.md Appendix B.

All arrays are float32 NCHW, parameters use the reference's state-dict key names.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblic_oracle.so")
_lib = None

PEDESTAL = float(2.0 ** -36)  # (2**-18)**2
BETA_MIN = 1e-6
LIKELIHOOD_BOUND = 1e-9
FE_NPARAM = 43


def build(force: bool = False) -> str:
    """Compile oracle/lic_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "lic_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_ci = ctypes.c_int
_cl = ctypes.c_long
_cf = ctypes.c_float


# ------------------------------------------------------------------------------------------
# op-level wrappers
# ------------------------------------------------------------------------------------------
def conv2d_fwd(x, w, b, stride, pad):
    x, w = _f(x), _f(w)
    B, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    y = np.empty((B, Co, Ho, Wo), np.float32)
    bb = None if b is None else _f(b)
    lib().lic_oracle_conv2d_fwd(_p(x), _p(w), _p(bb), _p(y), _ci(B), _ci(Ci), _ci(H), _ci(W),
                                _ci(Co), _ci(kh), _ci(kw), _ci(stride), _ci(pad))
    return y


def conv2d_bwd(x, w, dy, stride, pad, need_dx=True):
    x, w, dy = _f(x), _f(w), _f(dy)
    B, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty((Co,), np.float32)
    lib().lic_oracle_conv2d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), _ci(B), _ci(Ci),
                                _ci(H), _ci(W), _ci(Co), _ci(kh), _ci(kw), _ci(stride), _ci(pad))
    return dx, dw, db


def convT2d_fwd(x, w, b, stride, pad, out_pad):
    x, w = _f(x), _f(w)
    B, Ci, H, W = x.shape
    _, Co, kh, kw = w.shape
    Ho = (H - 1) * stride - 2 * pad + kh + out_pad
    Wo = (W - 1) * stride - 2 * pad + kw + out_pad
    y = np.empty((B, Co, Ho, Wo), np.float32)
    bb = None if b is None else _f(b)
    lib().lic_oracle_convT2d_fwd(_p(x), _p(w), _p(bb), _p(y), _ci(B), _ci(Ci), _ci(H), _ci(W),
                                 _ci(Co), _ci(kh), _ci(kw), _ci(stride), _ci(pad), _ci(out_pad))
    return y


def convT2d_bwd(x, w, dy, stride, pad, out_pad, need_dx=True):
    x, w, dy = _f(x), _f(w), _f(dy)
    B, Ci, H, W = x.shape
    _, Co, kh, kw = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty((Co,), np.float32)
    lib().lic_oracle_convT2d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), _ci(B), _ci(Ci),
                                 _ci(H), _ci(W), _ci(Co), _ci(kh), _ci(kw), _ci(stride), _ci(pad),
                                 _ci(out_pad))
    return dx, dw, db


def leaky_relu_fwd(x, slope=0.01):
    x = _f(x)
    y = np.empty_like(x)
    lib().lic_oracle_leaky_relu_fwd(_p(x), _p(y), _cl(x.size), _cf(slope))
    return y


def leaky_relu_bwd(y, dy, slope=0.01):
    y, dy = _f(y), _f(dy)
    dx = np.empty_like(y)
    lib().lic_oracle_leaky_relu_bwd(_p(y), _p(dy), _p(dx), _cl(y.size), _cf(slope))
    return dx


def gdn_bounds(minimum):
    return float((minimum + PEDESTAL) ** 0.5)


def gdn_reparam(p, minimum):
    p = _f(p)
    out = np.empty_like(p)
    lib().lic_oracle_gdn_reparam(_p(p), _p(out), _cl(p.size), _cf(gdn_bounds(minimum)), _cf(PEDESTAL))
    return out


def gdn_reparam_bwd(p, dout, minimum):
    p, dout = _f(p), _f(dout)
    dp = np.empty_like(p)
    lib().lic_oracle_gdn_reparam_bwd(_p(p), _p(dout), _p(dp), _cl(p.size), _cf(gdn_bounds(minimum)))
    return dp


def gdn_fwd(x, beta_e, gamma_e, inverse):
    x, beta_e, gamma_e = _f(x), _f(beta_e), _f(gamma_e)
    B, C, H, W = x.shape
    y = np.empty_like(x)
    norm = np.empty_like(x)
    lib().lic_oracle_gdn_fwd(_p(x), _p(beta_e), _p(gamma_e), _p(y), _p(norm), _ci(B), _ci(C),
                             _cl(H * W), _ci(int(inverse)))
    return y, norm


def gdn_bwd(x, norm, gamma_e, dy, inverse):
    x, norm, gamma_e, dy = _f(x), _f(norm), _f(gamma_e), _f(dy)
    B, C, H, W = x.shape
    dx = np.empty_like(x)
    dbeta = np.empty((C,), np.float32)
    dgamma = np.empty((C, C), np.float32)
    lib().lic_oracle_gdn_bwd(_p(x), _p(norm), _p(gamma_e), _p(dy), _p(dx), _p(dbeta), _p(dgamma),
                             _ci(B), _ci(C), _cl(H * W), _ci(int(inverse)))
    return dx, dbeta, dgamma


def entropy_params_fwd(raw, M, K):
    raw = _f(raw)
    B, _, H, W = raw.shape
    out = np.empty_like(raw)
    lib().lic_oracle_entropy_params_fwd(_p(raw), _p(out), _ci(B), _ci(M), _ci(K), _cl(H * W))
    return out


def entropy_params_bwd(raw, out, dout, M, K):
    raw, out, dout = _f(raw), _f(out), _f(dout)
    B, _, H, W = raw.shape
    draw = np.empty_like(raw)
    lib().lic_oracle_entropy_params_bwd(_p(raw), _p(out), _p(dout), _p(draw), _ci(B), _ci(M), _ci(K),
                                        _cl(H * W))
    return draw


def gmm_likelihood_fwd(x, params, K, bound=LIKELIHOOD_BOUND):
    x, params = _f(x), _f(params)
    B, M, H, W = x.shape
    p = np.empty_like(x)
    logp = np.empty_like(x)
    lib().lic_oracle_gmm_likelihood_fwd(_p(x), _p(params), _p(p), _p(logp), _ci(B), _ci(M), _ci(K),
                                        _cl(H * W), _cf(bound))
    return p, logp


def gmm_likelihood_bwd(x, params, dp, dlogp, K, bound=LIKELIHOOD_BOUND):
    x, params = _f(x), _f(params)
    B, M, H, W = x.shape
    dp = None if dp is None else _f(dp)
    dlogp = None if dlogp is None else _f(dlogp)
    dx = np.empty_like(x)
    dparams = np.empty_like(params)
    lib().lic_oracle_gmm_likelihood_bwd(_p(x), _p(params), _p(dp), _p(dlogp), _p(dx), _p(dparams),
                                        _ci(B), _ci(M), _ci(K), _cl(H * W), _cf(bound))
    return dx, dparams


# factorised-bottleneck parameter packing: state-dict tensors <-> [C][43]
_FE_SHAPES_M = [(3, 1), (3, 3), (3, 3), (1, 3)]
_FE_SHAPES_B = [(3, 1), (3, 1), (3, 1), (1, 1)]
_FE_SHAPES_F = [(3, 1), (3, 1), (3, 1)]


def fe_pack(matrices, biases, factors):
    C = matrices[0].shape[0]
    cols = [np.asarray(m, np.float32).reshape(C, -1) for m in matrices]
    cols += [np.asarray(b, np.float32).reshape(C, -1) for b in biases]
    cols += [np.asarray(f, np.float32).reshape(C, -1) for f in factors]
    out = np.concatenate(cols, axis=1)
    assert out.shape == (C, FE_NPARAM)
    return np.ascontiguousarray(out)


def fe_unpack(packed):
    C = packed.shape[0]
    o = 0
    mats, bs, fs = [], [], []
    for shp in _FE_SHAPES_M:
        n = shp[0] * shp[1]
        mats.append(packed[:, o:o + n].reshape(C, *shp).copy())
        o += n
    for shp in _FE_SHAPES_B:
        n = shp[0] * shp[1]
        bs.append(packed[:, o:o + n].reshape(C, *shp).copy())
        o += n
    for shp in _FE_SHAPES_F:
        n = shp[0] * shp[1]
        fs.append(packed[:, o:o + n].reshape(C, *shp).copy())
        o += n
    return mats, bs, fs


def factorized_fwd(x, packed, bound=LIKELIHOOD_BOUND):
    x, packed = _f(x), _f(packed)
    B, C = x.shape[:2]
    HW = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    p = np.empty_like(x)
    logp = np.empty_like(x)
    lib().lic_oracle_factorized_fwd(_p(x), _p(packed), _p(p), _p(logp), _ci(B), _ci(C), _cl(HW),
                                    _cf(bound))
    return p, logp


def factorized_bwd(x, packed, dp, dlogp, bound=LIKELIHOOD_BOUND):
    x, packed = _f(x), _f(packed)
    B, C = x.shape[:2]
    HW = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    dp = None if dp is None else _f(dp)
    dlogp = None if dlogp is None else _f(dlogp)
    dx = np.empty_like(x)
    dparams = np.empty_like(packed)
    lib().lic_oracle_factorized_bwd(_p(x), _p(packed), _p(dp), _p(dlogp), _p(dx), _p(dparams),
                                    _ci(B), _ci(C), _cl(HW), _cf(bound))
    return dx, dparams


def factorized_channel_logits(packed, ch, xs):
    packed, xs = _f(packed), _f(xs)
    out = np.empty_like(xs)
    lib().lic_oracle_factorized_channel_logits(_p(packed), _ci(ch), _p(xs), _p(out), _cl(xs.size))
    return out


def quantize(v, u, training):
    v = _f(v)
    uu = None if u is None else _f(u)
    out = np.empty_like(v)
    lib().lic_oracle_quantize(_p(v), _p(uu), _p(out), _cl(v.size), _ci(int(training)))
    return out


def rd_loss_fwd(logp_y, logp_z, x_hat, x, lambda_rd):
    logp_y, logp_z, x_hat, x = _f(logp_y), _f(logp_z), _f(x_hat), _f(x)
    B = x.shape[0]
    out = np.zeros((11,), np.float32)
    mse_img = np.empty((B,), np.float32)
    psnr_img = np.empty((B,), np.float32)
    lib().lic_oracle_rd_loss_fwd(_p(logp_y), _cl(logp_y.size // B), _p(logp_z), _cl(logp_z.size // B),
                                 _p(x_hat), _p(x), _cl(x.size // B), _ci(B),
                                 _cl(x.shape[2] * x.shape[3]), _cf(lambda_rd), _p(out), _p(mse_img),
                                 _p(psnr_img))
    keys = ["loss", "bpp_y", "bpp_z", "bpp_total", "mse", "psnr", "bits_y", "bits_z", "bits_total"]
    res = {k: float(out[i]) for i, k in enumerate(keys)}
    res["mse_per_image"] = mse_img
    res["psnr_per_image"] = psnr_img
    return res


def rd_loss_bwd(logp_y, logp_z, x_hat, x, lambda_rd, gl=1.0):
    x_hat, x = _f(x_hat), _f(x)
    B = x.shape[0]
    dly = np.empty(logp_y.shape, np.float32)
    dlz = np.empty(logp_z.shape, np.float32)
    dxh = np.empty_like(x_hat)
    lib().lic_oracle_rd_loss_bwd(_p(x_hat), _p(x), _cl(dly.size // B), _cl(dlz.size // B),
                                 _cl(x.size // B), _ci(B), _cl(x.shape[2] * x.shape[3]),
                                 _cf(lambda_rd), _cf(gl), _p(dly), _p(dlz), _p(dxh))
    return dly, dlz, dxh


# ------------------------------------------------------------------------------------------
# a minimal reverse-mode tape over the ops above
# ------------------------------------------------------------------------------------------
class V:
    __slots__ = ("d", "g")

    def __init__(self, d):
        self.d = d
        self.g = None

    def acc(self, g):
        self.g = g.copy() if self.g is None else self.g + g


class Tape:
    def __init__(self, params: Dict[str, np.ndarray]):
        self.params = {k: np.ascontiguousarray(v, np.float32) for k, v in params.items()
                       if np.asarray(v).dtype.kind == "f"}
        self.pgrad: Dict[str, np.ndarray] = {}
        self.fns = []

    def pacc(self, key, g):
        g = np.asarray(g, np.float32).reshape(self.params[key].shape)
        self.pgrad[key] = g.copy() if key not in self.pgrad else self.pgrad[key] + g

    def backward(self):
        for fn in reversed(self.fns):
            fn()

    # -- layers ---------------------------------------------------------------------------
    def conv(self, x: V, prefix, stride, pad, need_dx=True, weight_override=None):
        w = self.params[prefix + ".weight"] if weight_override is None else weight_override
        b = self.params[prefix + ".bias"]
        y = V(conv2d_fwd(x.d, w, b, stride, pad))

        def bw():
            if y.g is None:
                return
            dx, dw, db = conv2d_bwd(x.d, w, y.g, stride, pad, need_dx)
            if need_dx:
                x.acc(dx)
            self.pacc(prefix + ".weight", dw)
            self.pacc(prefix + ".bias", db)
        self.fns.append(bw)
        return y

    def convT(self, x: V, prefix, stride, pad, out_pad):
        w = self.params[prefix + ".weight"]
        b = self.params[prefix + ".bias"]
        y = V(convT2d_fwd(x.d, w, b, stride, pad, out_pad))

        def bw():
            if y.g is None:
                return
            dx, dw, db = convT2d_bwd(x.d, w, y.g, stride, pad, out_pad)
            x.acc(dx)
            self.pacc(prefix + ".weight", dw)
            self.pacc(prefix + ".bias", db)
        self.fns.append(bw)
        return y

    def leaky(self, x: V):
        y = V(leaky_relu_fwd(x.d))

        def bw():
            if y.g is not None:
                x.acc(leaky_relu_bwd(y.d, y.g))
        self.fns.append(bw)
        return y

    def gdn(self, x: V, prefix, inverse):
        beta_p = self.params[prefix + ".beta"]
        gamma_p = self.params[prefix + ".gamma"]
        beta_e = gdn_reparam(beta_p, BETA_MIN)
        gamma_e = gdn_reparam(gamma_p, 0.0)
        yd, norm = gdn_fwd(x.d, beta_e, gamma_e, inverse)
        y = V(yd)

        def bw():
            if y.g is None:
                return
            dx, dbe, dge = gdn_bwd(x.d, norm, gamma_e, y.g, inverse)
            x.acc(dx)
            self.pacc(prefix + ".beta", gdn_reparam_bwd(beta_p, dbe, BETA_MIN))
            self.pacc(prefix + ".gamma", gdn_reparam_bwd(gamma_p, dge, 0.0))
        self.fns.append(bw)
        return y

    def add(self, a: V, b: V):
        y = V(a.d + b.d)

        def bw():
            if y.g is not None:
                a.acc(y.g)
                b.acc(y.g)
        self.fns.append(bw)
        return y

    def cat(self, a: V, b: V):
        ca = a.d.shape[1]
        y = V(np.concatenate([a.d, b.d], axis=1))

        def bw():
            if y.g is not None:
                a.acc(np.ascontiguousarray(y.g[:, :ca]))
                b.acc(np.ascontiguousarray(y.g[:, ca:]))
        self.fns.append(bw)
        return y


# ------------------------------------------------------------------------------------------
# component stacks (Components.py / Layers.py)
# ------------------------------------------------------------------------------------------
def _rb_stride(t: Tape, x: V, pre, stride=2, first=False):
    """ResidualBlockWithStride (Layers.py:27-58)."""
    out = t.conv(x, pre + ".conv1", stride, 1, need_dx=not first)
    out = t.leaky(out)
    out = t.conv(out, pre + ".conv2", 1, 1)
    out = t.gdn(out, pre + ".gdn", False)
    if (pre + ".skip.weight") in t.params:
        idn = t.conv(x, pre + ".skip", stride, 0, need_dx=not first)
    else:
        idn = x
    return t.add(out, idn)


def _rb(t: Tape, x: V, pre):
    """ResidualBlock (Layers.py:89-119)."""
    out = t.conv(x, pre + ".conv1", 1, 1)
    out = t.leaky(out)
    out = t.conv(out, pre + ".conv2", 1, 1)
    out = t.leaky(out)
    if (pre + ".skip.weight") in t.params:
        idn = t.conv(x, pre + ".skip", 1, 0)
    else:
        idn = x
    return t.add(out, idn)


def _rb_up(t: Tape, x: V, pre):
    """ResidualBlockUpsample (Layers.py:61-86) with TransposedDeconv3x3 (Layers.py:18-24)."""
    out = t.convT(x, pre + ".subpel_conv.deconv", 2, 1, 1)
    out = t.leaky(out)
    out = t.conv(out, pre + ".conv", 1, 1)
    out = t.gdn(out, pre + ".igdn", True)
    idn = t.convT(x, pre + ".upsample.deconv", 2, 1, 1)
    return t.add(out, idn)


def encoder(t: Tape, x: V, kind):
    p = "encoder.net."
    if kind == "5x5":  # Components.py:6-18
        h = t.conv(x, p + "0", 2, 2, need_dx=False)
        h = t.gdn(h, p + "1", False)
        h = t.conv(h, p + "2", 2, 2)
        h = t.gdn(h, p + "3", False)
        h = t.conv(h, p + "4", 2, 2)
        h = t.gdn(h, p + "5", False)
        return t.conv(h, p + "6", 2, 2)
    # Components.py:20-32
    h = _rb_stride(t, x, p + "0", 2, first=True)
    h = _rb(t, h, p + "1")
    h = _rb_stride(t, h, p + "2", 2)
    h = _rb(t, h, p + "3")
    h = _rb_stride(t, h, p + "4", 2)
    h = _rb(t, h, p + "5")
    return t.conv(h, p + "6", 2, 1)


def decoder(t: Tape, y: V, kind):
    p = "decoder.net."
    if kind == "5x5":  # Components.py:35-47
        h = t.convT(y, p + "0", 2, 2, 1)
        h = t.gdn(h, p + "1", True)
        h = t.convT(h, p + "2", 2, 2, 1)
        h = t.gdn(h, p + "3", True)
        h = t.convT(h, p + "4", 2, 2, 1)
        h = t.gdn(h, p + "5", True)
        return t.convT(h, p + "6", 2, 2, 1)
    # Components.py:49-62
    h = _rb(t, y, p + "0")
    h = _rb_up(t, h, p + "1")
    h = _rb(t, h, p + "2")
    h = _rb_up(t, h, p + "3")
    h = _rb(t, h, p + "4")
    h = _rb_up(t, h, p + "5")
    h = _rb(t, h, p + "6")
    return t.convT(h, p + "7.deconv", 2, 1, 1)


def hyper_encoder(t: Tape, y: V, kind):
    p = "hyper_encoder.net."
    if kind == "5x5":  # Components.py:65-75
        h = t.leaky(t.conv(y, p + "0", 1, 1))
        h = t.leaky(t.conv(h, p + "2", 2, 2))
        return t.conv(h, p + "4", 2, 2)
    # Components.py:77-91
    h = t.leaky(t.conv(y, p + "0", 1, 1))
    h = t.leaky(t.conv(h, p + "2", 1, 1))
    h = t.leaky(t.conv(h, p + "4", 2, 1))
    h = t.leaky(t.conv(h, p + "6", 1, 1))
    return t.conv(h, p + "8", 2, 1)


def hyper_decoder(t: Tape, z: V, kind):
    p = "hyper_decoder.net."
    if kind == "5x5":  # Components.py:94-105
        h = t.leaky(t.convT(z, p + "0", 2, 2, 1))
        h = t.leaky(t.convT(h, p + "2", 2, 2, 1))
        return t.conv(h, p + "4", 1, 1)
    # Components.py:107-122
    h = t.leaky(t.conv(z, p + "0", 1, 1))
    h = t.leaky(t.convT(h, p + "2.deconv", 2, 1, 1))
    h = t.leaky(t.conv(h, p + "4", 1, 1))
    h = t.leaky(t.convT(h, p + "6.deconv", 2, 1, 1))
    return t.conv(h, p + "8", 1, 1)


def mask_a(shape):
    """Type-A mask (ContextModels.py:12-16)."""
    m = np.ones(shape, np.float32)
    kH, kW = shape[2], shape[3]
    m[:, :, kH // 2, kW // 2:] = 0
    m[:, :, kH // 2 + 1:] = 0
    return m


def model_forward(params: Dict[str, np.ndarray], x: np.ndarray, M: int, K: int, kind: str = "5x5",
                  training: bool = True, noise: Optional[Tuple[np.ndarray, np.ndarray]] = None,
                  lambda_rd: Optional[float] = None, backward: bool = False):
    """JointAutoregressiveHierarchical (kind='5x5', Models.py:49-106) or
    HierarchicalMixtureResidual (kind='3x3', Models.py:148-205) forward; optional rd_loss and
    full backward.  Returns (out_dict, loss_dict|None, param_grads|None).
    `noise` = (u_z, u_y) uniform [0,1) samples in the reference's draw order (z first)."""
    t = Tape(params)
    xv = V(_f(x))
    y = encoder(t, xv, kind)
    z = hyper_encoder(t, y, kind)
    if training:
        uz, uy = noise
        z_in = V(quantize(z.d, uz, True))
        y_in = V(quantize(y.d, uy, True))

        def bw_q():
            if z_in.g is not None:
                z.acc(z_in.g)
            if y_in.g is not None:
                y.acc(y_in.g)
        t.fns.append(bw_q)
    else:
        z_in = V(quantize(z.d, None, False))
        y_in = V(quantize(y.d, None, False))
    psi = hyper_decoder(t, z_in, kind)
    # masked conv: weight.data *= mask in place, ordinary conv; weight grads are NOT masked
    wkey = "context_model.masked.weight"
    wm = t.params[wkey] * mask_a(t.params[wkey].shape)
    t.params[wkey] = wm  # in-place semantics (ContextModels.py:19)
    phi = t.conv(y_in, "context_model.masked", 1, 2, weight_override=wm)
    comb = t.cat(phi, psi)
    h = t.leaky(t.conv(comb, "entropy_parameters.net.0", 1, 0))
    h = t.leaky(t.conv(h, "entropy_parameters.net.2", 1, 0))
    raw = t.conv(h, "entropy_parameters.net.4", 1, 0)
    ep = V(entropy_params_fwd(raw.d, M, K))

    def bw_ep():
        if ep.g is not None:
            raw.acc(entropy_params_bwd(raw.d, ep.d, ep.g, M, K))
    t.fns.append(bw_ep)

    fe_keys = ([f"factorized_entropy_model.matrices.{i}" for i in range(4)],
               [f"factorized_entropy_model.biases.{i}" for i in range(4)],
               [f"factorized_entropy_model.factors.{i}" for i in range(3)])
    packed = fe_pack([t.params[k] for k in fe_keys[0]], [t.params[k] for k in fe_keys[1]],
                     [t.params[k] for k in fe_keys[2]])
    pz, lpz = factorized_fwd(z_in.d, packed)
    p_z, logp_z = V(pz), V(lpz)

    def bw_fe():
        if p_z.g is None and logp_z.g is None:
            return
        dx, dpk = factorized_bwd(z_in.d, packed, p_z.g, logp_z.g)
        z_in.acc(dx)
        mats, bs, fs = fe_unpack(dpk)
        for k, g in zip(fe_keys[0] + fe_keys[1] + fe_keys[2], mats + bs + fs):
            t.pacc(k, g)
    t.fns.append(bw_fe)

    py, lpy = gmm_likelihood_fwd(y_in.d, ep.d, K)
    p_y, logp_y = V(py), V(lpy)

    def bw_gm():
        if p_y.g is None and logp_y.g is None:
            return
        dx, dpar = gmm_likelihood_bwd(y_in.d, ep.d, p_y.g, logp_y.g, K)
        y_in.acc(dx)
        ep.acc(dpar)
    t.fns.append(bw_gm)

    x_hat = decoder(t, y_in, kind)

    B, _, h_, w_ = y.d.shape
    out = {"x_hat": x_hat.d, "y": y.d, "y_in": y_in.d, "z": z.d, "z_in": z_in.d, "p_z": p_z.d,
           "logp_z": logp_z.d, "p_y": p_y.d, "logp_y": logp_y.d, "training": training}
    if K == 1:
        out["mu"], out["sigma"] = ep.d[:, :M], ep.d[:, M:]
    else:
        third = K * M
        out["weights"] = ep.d[:, :third].reshape(B, K, M, h_, w_)
        out["mus"] = ep.d[:, third:2 * third].reshape(B, K, M, h_, w_)
        out["sigmas"] = ep.d[:, 2 * third:].reshape(B, K, M, h_, w_)
    loss = None
    grads = None
    if lambda_rd is not None:
        loss = rd_loss_fwd(logp_y.d, logp_z.d, x_hat.d, xv.d, lambda_rd)
        if backward:
            dly, dlz, dxh = rd_loss_bwd(logp_y.d, logp_z.d, x_hat.d, xv.d, lambda_rd, 1.0)
            logp_y.acc(dly)
            logp_z.acc(dlz)
            x_hat.acc(dxh)
            t.backward()
            grads = t.pgrad
            grads["__y"] = y.g
            grads["__z"] = z.g
    return out, loss, grads
