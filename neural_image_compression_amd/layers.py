"""Layer classes with the reference's names, constructor signatures, parameter shapes and
state-dict keys, executing on the gfx950 kernels (functional.py).

Mirrors (reference paths): Layers.py:6-119 (SubpelConv3x3 is never instantiated there and is
not provided), third-party compressai.layers.gdn.GDN as used at Components.py:11-44 and
Layers.py:41,75 (definition: SURVEY.md Appendix B; parity unpinned).

Conv2d / ConvTranspose2d subclass torch's modules only to inherit their parameter containers
and default initialisation (so `torch.manual_seed(s); Model()` draws the same weights as the
reference); forward never touches ATen's convolution.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from . import functional as F_
from . import functional_bf16 as FB_


def _pair(v):
    return v if isinstance(v, int) else v[0]


class Conv2d(nn.Conv2d):
    def forward(self, x: Tensor, leaky: bool = False, slope: float = 0.01, residual=None, bf16: bool = False,
                out_f32: bool = False, out=None) -> Tensor:
        if self.groups != 1 or _pair(self.dilation) != 1:
            raise NotImplementedError("groups/dilation are not used by the reference models")
        s, p = _pair(self.stride), _pair(self.padding)
        if out is not None and self.in_channels < 4:
            raise NotImplementedError("`out` slices are for the implicit-GEMM path")
        if bf16:  # bf16-storage path (BASELINE config 3): conv + bias (+ fused LeakyReLU)
            if residual is not None:
                raise NotImplementedError("bf16 mode covers the 5x5 conv/GDN stacks and the latent-side layers")
            if self.in_channels < 4:
                y = FB_.image_conv2d_bf16(x, self.weight, self.bias, s, p)
                # (the 3x3 model's first block: the activation of an RGB layer is an element-wise launch of its own)
                return torch.nn.functional.leaky_relu(y, slope) if leaky else y
            return FB_.conv2d_bf16(x, self.weight, self.bias, s, p, out_f32, leaky, slope, 0, out)
        if self.in_channels < 4:  # RGB stem: im2col + dense MFMA GEMM
            y = F_.image_conv2d(x, self.weight, self.bias, s, p, leaky, slope)
            return y if residual is None else y + residual
        return F_.conv2d(x, self.weight, self.bias, s, p, leaky, slope, 0, residual, out)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x: Tensor, leaky: bool = False, slope: float = 0.01, residual=None, bf16: bool = False,
                out_f32: bool = False) -> Tensor:
        if self.groups != 1 or _pair(self.dilation) != 1:
            raise NotImplementedError("groups/dilation are not used by the reference models")
        s, p, op = _pair(self.stride), _pair(self.padding), _pair(self.output_padding)
        if bf16:
            if residual is not None:
                raise NotImplementedError("bf16 mode covers the 5x5 conv/GDN stacks and the latent-side layers")
            if self.out_channels < 4:
                if leaky:
                    raise NotImplementedError
                return FB_.image_conv_transpose2d_bf16(x, self.weight, self.bias, s, p, op)
            return FB_.conv_transpose2d_bf16(x, self.weight, self.bias, s, p, op, out_f32, leaky, slope)
        if self.out_channels < 4:  # RGB head: dense MFMA GEMM + col2im
            if leaky or residual is not None:
                raise NotImplementedError
            return F_.image_conv_transpose2d(x, self.weight, self.bias, s, p, op)
        return F_.conv_transpose2d(x, self.weight, self.bias, s, p, op, leaky, slope, residual)


class LeakyReLU(nn.Module):
    """nn.LeakyReLU stand-in; inside the stacks it is fused into the preceding conv's epilogue."""

    def __init__(self, negative_slope: float = 0.01, inplace: bool = False):
        super().__init__()
        self.negative_slope = float(negative_slope)
        self.inplace = inplace

    def forward(self, x: Tensor) -> Tensor:
        return F_.leaky_relu(x, self.negative_slope)


def run_bf16(seq: nn.Sequential, x: Tensor, out_f32: bool = True, out=None) -> Tensor:
    """bf16-storage execution of a stack: bf16 between layers (Conv -> LeakyReLU pairs as one launch), the
    last layer writes fp32 when `out_f32` (latents, entropy parameters, the image) and bf16 otherwise
    (features that only feed another bf16 layer)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(m, (Conv2d, ConvTranspose2d)):
            if isinstance(nxt, LeakyReLU):
                x = m(x, bf16=True, leaky=True, slope=nxt.negative_slope)
                i += 2
                continue
            if FUSE_CONV_GDN and isinstance(nxt, GDN) and m.groups == 1 and _pair(m.dilation) == 1 and x.dim() == 4 and \
                    (isinstance(m, Conv2d) or m.out_channels >= 4) and \
                    FB_.fused_gdn_supported_bf16(max(m.in_channels, 8), m.out_channels):
                tr = isinstance(m, ConvTranspose2d)
                x = FB_.conv_gdn_bf16(x, m.weight, m.bias, nxt.beta, nxt.gamma, _pair(m.stride), _pair(m.padding),
                                     nxt.inverse, nxt.beta_reparam.bound_value, nxt.gamma_reparam.bound_value,
                                     nxt.beta_reparam.pedestal_value, transposed=tr,
                                     output_padding=_pair(m.output_padding) if tr else 0)
                i += 2
                continue
            if out is not None and i == len(mods) - 1:   # the stack's result goes into the caller's channel slice
                x = m(x, bf16=True, out=out)
            else:
                x = m(x, bf16=True, out_f32=(out_f32 and i == len(mods) - 1))
        elif isinstance(m, GDN):
            x = m(x, bf16=True)
        elif isinstance(m, TransposedDeconv3x3):
            last = out_f32 and i == len(mods) - 1
            if isinstance(nxt, LeakyReLU):
                x = m.deconv(x, bf16=True, leaky=True, slope=nxt.negative_slope)
                i += 2
                continue
            x = m.deconv(x, bf16=True, out_f32=last) if m.deconv.out_channels >= 4 else m.deconv(x, bf16=True)
        elif isinstance(m, (ResidualBlock, ResidualBlockWithStride, ResidualBlockUpsample)):
            x = m(x, bf16=True)
        else:
            raise NotImplementedError(f"bf16 mode does not cover {type(m).__name__}")
        i += 1
    return x.float() if (out_f32 and x.dtype != torch.float32) else x


# conv -> GDN pairs as one launch: "auto" = where it is measured to pay (layers on 64-row tiles),
# True = wherever the kernel supports it, False = never (tests flip it to compare the two paths)
FUSE_CONV_GDN = "auto"


def _conv_gdn(conv, g: "GDN", x: Tensor) -> Tensor:
    tr = isinstance(conv, ConvTranspose2d)
    return F_.conv_gdn(x, conv.weight, conv.bias, g.beta, g.gamma, _pair(conv.stride), _pair(conv.padding),
                       g.inverse, g.beta_reparam.bound_value, g.gamma_reparam.bound_value,
                       g.beta_reparam.pedestal_value, transposed=tr,
                       output_padding=_pair(conv.output_padding) if tr else 0)


def run_fused(seq: nn.Sequential, x: Tensor, out=None) -> Tensor:
    """Run an nn.Sequential of our layers, folding Conv -> LeakyReLU and Conv -> GDN pairs into one
    kernel each.  `out`: NHWC channel slice the LAST layer (a plain Conv2d) writes its result into."""
    mods = list(seq)
    if out is not None and not (isinstance(mods[-1], Conv2d) and mods[-1].in_channels >= 4):
        raise NotImplementedError("`out` needs a plain Conv2d as the last layer")
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        fusable = isinstance(m, Conv2d) or (isinstance(m, ConvTranspose2d) and m.out_channels >= 4) or \
            (type(m).__name__ == "TransposedDeconv3x3" and m.deconv.out_channels >= 4)
        if FUSE_CONV_GDN and isinstance(nxt, GDN) and isinstance(m, (Conv2d, ConvTranspose2d)) and \
                m.groups == 1 and _pair(m.dilation) == 1 and x.dim() == 4 and \
                F_.fused_gdn_supported(m.in_channels if m.in_channels >= 4 else 4, m.out_channels) and \
                (FUSE_CONV_GDN is True or F_.fused_gdn_preferred(
                    tuple(x.shape), tuple(m.weight.shape), _pair(m.stride), _pair(m.padding),
                    isinstance(m, ConvTranspose2d),
                    _pair(m.output_padding) if isinstance(m, ConvTranspose2d) else 0)):
            x = _conv_gdn(m, nxt, x)
            i += 2
        elif fusable and isinstance(nxt, LeakyReLU):
            x = m(x, leaky=True, slope=nxt.negative_slope)
            i += 2
        elif out is not None and i == len(mods) - 1:
            x = m(x, out=out)
            i += 1
        else:
            x = m(x)
            i += 1
    return x


# ---- GDN (compressai definition) -------------------------------------------------------------
class LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum: float = 0.0, reparam_offset: float = 2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((self.minimum + pedestal) ** 0.5)
        self.bound_value = float((self.minimum + pedestal) ** 0.5)
        self.pedestal_value = float(pedestal)

    def init(self, x: Tensor) -> Tensor:
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))


class GDN(nn.Module):
    """norm_i = beta_i + sum_j gamma_ij x_j^2;  y = x * rsqrt(norm)  (inverse: x * sqrt(norm))."""

    def __init__(self, in_channels: int, inverse: bool = False, beta_min: float = 1e-6,
                 gamma_init: float = 0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(int(in_channels))))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(int(in_channels))))

    def forward(self, x: Tensor, residual=None, bf16: bool = False) -> Tensor:
        if bf16:
            if residual is not None:
                raise NotImplementedError
            return FB_.gdn_bf16(x, self.beta, self.gamma, self.inverse, self.beta_reparam.bound_value,
                                self.gamma_reparam.bound_value, self.beta_reparam.pedestal_value)
        return F_.gdn(x, self.beta, self.gamma, self.inverse, self.beta_reparam.bound_value,
                      self.gamma_reparam.bound_value, self.beta_reparam.pedestal_value, residual)


def _conv_gdn_bf16(conv, g: "GDN", x: Tensor) -> Tensor:
    """conv -> GDN / IGDN of a residual block in bf16 storage: one launch where the fused kernel covers the pair"""
    if FUSE_CONV_GDN and x.dim() == 4 and FB_.fused_gdn_supported_bf16(max(conv.in_channels, 8), conv.out_channels):
        return FB_.conv_gdn_bf16(x, conv.weight, conv.bias, g.beta, g.gamma, _pair(conv.stride), _pair(conv.padding),
                                 g.inverse, g.beta_reparam.bound_value, g.gamma_reparam.bound_value,
                                 g.beta_reparam.pedestal_value)
    return g(conv(x, bf16=True), bf16=True)


# ---- residual blocks (Layers.py:18-119) --------------------------------------------------------
class TransposedDeconv3x3(nn.Module):
    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.deconv = ConvTranspose2d(in_ch, out_ch, kernel_size=3, stride=upsample, padding=1,
                                      output_padding=upsample - 1)

    def forward(self, x, leaky=False, slope=0.01):
        return self.deconv(x, leaky=leaky, slope=slope)


class ResidualBlockWithStride(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, stride: int = 2):
        super().__init__()
        self.conv1 = Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)
        self.leaky_relu = LeakyReLU(inplace=True)
        self.conv2 = Conv2d(out_ch, out_ch, kernel_size=3, stride=1, padding=1)
        self.gdn = GDN(out_ch, beta_min=1e-6, gamma_init=.1)
        self.skip = Conv2d(in_ch, out_ch, kernel_size=1, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def forward(self, x: Tensor, bf16: bool = False) -> Tensor:
        if bf16:   # bf16 storage: the same graph, the skip connection as a bf16 add behind the (fused) conv + GDN
            out = self.conv1(x, bf16=True, leaky=True, slope=self.leaky_relu.negative_slope)
            out = _conv_gdn_bf16(self.conv2, self.gdn, out)
            identity = FB_.as_bf16(x) if self.skip is None else self.skip(x, bf16=True)
            return out + identity
        out = self.conv1(x, leaky=True, slope=self.leaky_relu.negative_slope)
        out = self.conv2(out)
        identity = x if self.skip is None else self.skip(x)
        return self.gdn(out, residual=identity)  # out += identity fused into the GDN epilogue


class ResidualBlockUpsample(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, upsample: int = 2):
        super().__init__()
        self.subpel_conv = TransposedDeconv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = LeakyReLU(inplace=True)
        self.conv = Conv2d(out_ch, out_ch, kernel_size=3, stride=1, padding=1)
        self.igdn = GDN(out_ch, inverse=True, beta_min=1e-6, gamma_init=.1)
        self.upsample = TransposedDeconv3x3(in_ch, out_ch, upsample)

    def forward(self, x: Tensor, bf16: bool = False) -> Tensor:
        if bf16:
            out = self.subpel_conv.deconv(x, bf16=True, leaky=True, slope=self.leaky_relu.negative_slope)
            out = _conv_gdn_bf16(self.conv, self.igdn, out)
            return out + self.upsample.deconv(x, bf16=True)
        out = self.subpel_conv(x, leaky=True, slope=self.leaky_relu.negative_slope)
        out = self.conv(out)
        identity = self.upsample(x)
        return self.igdn(out, residual=identity)


class ResidualBlock(nn.Module):
    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.conv1 = Conv2d(in_ch, out_ch, kernel_size=3, stride=1, padding=1)
        self.leaky_relu = LeakyReLU(inplace=True)
        self.conv2 = Conv2d(out_ch, out_ch, kernel_size=3, stride=1, padding=1)
        self.skip = Conv2d(in_ch, out_ch, kernel_size=1, stride=1) if in_ch != out_ch else None

    def forward(self, x: Tensor, bf16: bool = False) -> Tensor:
        s = self.leaky_relu.negative_slope
        if bf16:
            out = self.conv2(self.conv1(x, bf16=True, leaky=True, slope=s), bf16=True, leaky=True, slope=s)
            return out + (FB_.as_bf16(x) if self.skip is None else self.skip(x, bf16=True))
        out = self.conv1(x, leaky=True, slope=s)
        identity = x if self.skip is None else self.skip(x)
        return self.conv2(out, leaky=True, slope=s, residual=identity)  # leaky(conv2) + identity
