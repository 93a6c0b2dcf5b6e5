"""Analysis / synthesis / hyper stacks with the reference's class names, constructor
signatures and `net.<i>` state-dict keys (reference: Components.py:6-122), and
`LatentSpaceTransform` (Components.py:125-153) with the one change that lets it execute (see the class)."""
from __future__ import annotations

import torch.nn as nn

from .layers import (Conv2d, ConvTranspose2d, GDN, LeakyReLU, ResidualBlock, ResidualBlockUpsample,
                     ResidualBlockWithStride, TransposedDeconv3x3, run_bf16, run_fused)


class _Stack(nn.Module):
    precision = "fp32"  # "bf16": bf16 activations between the layers (5x5 stacks and the hyper stacks)
    out_f32 = True      # bf16 mode: the stack's result in fp32 (latents, image) or bf16 (features for a bf16 consumer)

    def forward(self, x, out=None):
        if self.precision == "bf16":
            if out is not None and (self.out_f32 or not isinstance(self.net[-1], Conv2d)):
                raise NotImplementedError("`out` needs a stack that ends in a plain Conv2d and hands bf16 features on")
            return run_bf16(self.net, x, self.out_f32, out)
        return run_fused(self.net, x, out)


class Encoder5x5(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        self.net = nn.Sequential(
            Conv2d(3, M, kernel_size=5, stride=2, padding=2), GDN(M, beta_min=1e-6, gamma_init=.1),
            Conv2d(M, M, kernel_size=5, stride=2, padding=2), GDN(M, beta_min=1e-6, gamma_init=.1),
            Conv2d(M, M, kernel_size=5, stride=2, padding=2), GDN(M, beta_min=1e-6, gamma_init=.1),
            Conv2d(M, M, kernel_size=5, stride=2, padding=2),
        )


class Encoder3x3(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        self.net = nn.Sequential(
            ResidualBlockWithStride(3, M, stride=2), ResidualBlock(M, M),
            ResidualBlockWithStride(M, M, stride=2), ResidualBlock(M, M),
            ResidualBlockWithStride(M, M, stride=2), ResidualBlock(M, M),
            Conv2d(M, M, kernel_size=3, stride=2, padding=1),
        )


class Decoder5x5(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        ct = lambda ci, co: ConvTranspose2d(ci, co, kernel_size=5, stride=2, padding=2, output_padding=1)
        self.net = nn.Sequential(
            ct(M, M), GDN(M, inverse=True, beta_min=1e-6, gamma_init=.1),
            ct(M, M), GDN(M, inverse=True, beta_min=1e-6, gamma_init=.1),
            ct(M, M), GDN(M, inverse=True, beta_min=1e-6, gamma_init=.1),
            ct(M, 3),
        )


class Decoder3x3(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        self.net = nn.Sequential(
            ResidualBlock(M, M), ResidualBlockUpsample(M, M, 2),
            ResidualBlock(M, M), ResidualBlockUpsample(M, M, 2),
            ResidualBlock(M, M), ResidualBlockUpsample(M, M, 2),
            ResidualBlock(M, M), TransposedDeconv3x3(M, 3, 2),
        )


class HyperEncoder5x5(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        self.net = nn.Sequential(
            Conv2d(M, M, kernel_size=3, stride=1, padding=1), LeakyReLU(inplace=True),
            Conv2d(M, M, kernel_size=5, stride=2, padding=2), LeakyReLU(inplace=True),
            Conv2d(M, M, kernel_size=5, stride=2, padding=2),
        )


class HyperEncoder3x3(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        self.net = nn.Sequential(
            Conv2d(M, M, kernel_size=3, stride=1, padding=1), LeakyReLU(inplace=True),
            Conv2d(M, M, kernel_size=3, stride=1, padding=1), LeakyReLU(inplace=True),
            Conv2d(M, M, kernel_size=3, stride=2, padding=1), LeakyReLU(inplace=True),
            Conv2d(M, M, kernel_size=3, stride=1, padding=1), LeakyReLU(inplace=True),
            Conv2d(M, M, kernel_size=3, stride=2, padding=1),
        )


class HyperDecoder5x5(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        self.net = nn.Sequential(
            ConvTranspose2d(M, M, kernel_size=5, stride=2, padding=2, output_padding=1), LeakyReLU(inplace=True),
            ConvTranspose2d(M, int(1.5 * M), kernel_size=5, stride=2, padding=2, output_padding=1),
            LeakyReLU(inplace=True),
            Conv2d(int(1.5 * M), 2 * M, kernel_size=3, stride=1, padding=1),
        )


class HyperDecoder3x3(_Stack):
    def __init__(self, latent_channels=192):
        super().__init__()
        M = latent_channels
        self.net = nn.Sequential(
            Conv2d(M, M, kernel_size=3, stride=1, padding=1), LeakyReLU(inplace=True),
            TransposedDeconv3x3(M, M, 2), LeakyReLU(inplace=True),
            Conv2d(M, int(1.5 * M), kernel_size=3, stride=1, padding=1), LeakyReLU(inplace=True),
            TransposedDeconv3x3(int(1.5 * M), int(1.5 * M), 2), LeakyReLU(inplace=True),
            Conv2d(int(1.5 * M), 2 * M, kernel_size=3, stride=1, padding=1),
        )


class LatentSpaceTransform(nn.Module):
    """Components.py:125-153: RB -> URB(up u0) -> RB -> URB(u1) -> RB -> URB(u2) -> RB -> conv3x3(C -> C*u3), attribute
    names (state-dict keys) as the reference's.  ONE deliberate difference: the reference multiplies its CHANNEL
    count by each SPATIAL upsampling factor (`latent_channels *= upsampling_factors[i]`, Components.py:130,134,138),
    so with the only configuration it is used with, [2,1,1,1], `RB2` expects 2C channels, receives C and raises
     -- the reference's class cannot run.  Here every block keeps C channels (what
    `ResidualBlockUpsample(in_ch=C, out_ch=C)` produces), which is the smallest change that executes."""

    def __init__(self, latent_channels=192, upsampling_factors=(2, 1, 1, 1)):
        super().__init__()
        C, u = int(latent_channels), [int(v) for v in upsampling_factors]
        if len(u) != 4 or any(v < 1 or v > 2 for v in u[:3]):
            raise ValueError("upsampling_factors: four entries, spatial factors 1 or 2")
        self.RB1 = ResidualBlock(in_ch=C, out_ch=C)
        self.URB1 = ResidualBlockUpsample(in_ch=C, out_ch=C, upsample=u[0])
        self.RB2 = ResidualBlock(in_ch=C, out_ch=C)
        self.URB2 = ResidualBlockUpsample(in_ch=C, out_ch=C, upsample=u[1])
        self.RB3 = ResidualBlock(in_ch=C, out_ch=C)
        self.URB3 = ResidualBlockUpsample(in_ch=C, out_ch=C, upsample=u[2])
        self.RB4 = ResidualBlock(in_ch=C, out_ch=C)
        self.conv = Conv2d(C, C * u[3], kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        for blk in (self.RB1, self.URB1, self.RB2, self.URB2, self.RB3, self.URB3, self.RB4, self.conv):
            x = blk(x)
        return x
