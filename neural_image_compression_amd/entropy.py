"""Entropy models, context model and entropy-parameter network with the reference's class
names and state-dict keys (reference: EntropyModels.py:11-233, ContextModels.py:3-36,
ParametersModels.py:8-64, utils.py:6-8), executing on the gfx950 kernels."""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import functional as F_
from . import functional_bf16 as FB_
from .layers import Conv2d, LeakyReLU, run_bf16, run_fused


class EntropyModel(nn.Module):
    """Base interface (EntropyModels.py:11-46): forward = _likelihood(...).clamp_min(bound)."""

    def __init__(self, likelihood_lower_bound: float = 1e-9):
        super().__init__()
        self.likelihood_lower_bound = likelihood_lower_bound

    def _likelihood(self, inputs: Tensor, **kwargs) -> Tensor:
        raise NotImplementedError

    def forward(self, inputs: Tensor, **kwargs) -> Tensor:
        return self.likelihood_and_log(inputs, **kwargs)[0]

    def likelihood_and_log(self, inputs: Tensor, **kwargs) -> Tuple[Tensor, Tensor]:
        raise NotImplementedError

    def channel_cdf(self, ch: int, x: Tensor) -> Tensor:
        raise NotImplementedError

    def channel_pmf(self, ch: int, x: Tensor) -> Tensor:
        raise NotImplementedError

    @property
    def likelihood_bound(self) -> float:
        return self.likelihood_lower_bound


class FactorizedEntropyBottleneck(EntropyModel):
    """EntropyModels.py:49-184: per-channel 1-3-3-3-1 cumulative-logit network."""

    def __init__(self, channels: int, init_scale: float = 10.0, hidden_dims: Tuple[int, ...] = (3, 3, 3),
                 likelihood_lower_bound: float = 1e-9):
        super().__init__(likelihood_lower_bound)
        self.channels = int(channels)
        self.init_scale = float(init_scale)
        self.filters = tuple(int(f) for f in hidden_dims)
        if self.filters != (3, 3, 3):
            raise NotImplementedError("the fused kernel implements the reference's (3,3,3) network")
        self.dtype = torch.float32
        filters_full = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1.0 / (len(self.filters) + 1))
        self.matrices = nn.ParameterList()
        self.biases = nn.ParameterList()
        self.factors = nn.ParameterList()
        for i in range(len(self.filters) + 1):
            out, inp = filters_full[i + 1], filters_full[i]
            init_val = math.log(math.expm1(1.0 / scale / out))
            self.matrices.append(nn.Parameter(torch.full((self.channels, out, inp), init_val, dtype=self.dtype)))
            b = nn.Parameter(torch.empty((self.channels, out, 1), dtype=self.dtype))
            nn.init.uniform_(b, -0.5, 0.5)
            self.biases.append(b)
            if i < len(self.filters):
                self.factors.append(nn.Parameter(torch.zeros((self.channels, out, 1), dtype=self.dtype)))

    def _plists(self):
        return list(self.matrices), list(self.biases), list(self.factors)

    def packed_params(self) -> Tensor:
        """[C, 43]: matrices | biases | factors per channel, the layout of the lic_factorized_* entries."""
        Cc = self.channels
        return torch.cat([q.reshape(Cc, -1) for q in (*self.matrices, *self.biases, *self.factors)], dim=1).contiguous()

    def likelihood_and_log(self, inputs: Tensor, bound=None) -> Tuple[Tensor, Tensor]:
        if inputs.dim() < 2:
            raise ValueError("inputs must be at least 2D with channel axis")
        m, b, f = self._plists()
        return F_.factorized_likelihood(inputs, m, b, f,
                                        self.likelihood_lower_bound if bound is None else bound)

    def _likelihood(self, inputs: Tensor) -> Tensor:
        return self.likelihood_and_log(inputs, bound=0.0)[0]

    @torch.no_grad()
    def channel_logits_cumulative(self, ch: int, x: Tensor) -> Tensor:
        m, b, f = self._plists()
        return F_.factorized_channel_logits(m, b, f, ch, x.reshape(-1)).view(-1)

    @torch.no_grad()
    def channel_cdf(self, ch: int, x: Tensor) -> Tensor:
        return torch.sigmoid(self.channel_logits_cumulative(ch, x))

    @torch.no_grad()
    def channel_pmf(self, ch: int, x: Tensor) -> Tensor:
        Lp = self.channel_logits_cumulative(ch, x + 0.5)
        Lm = self.channel_logits_cumulative(ch, x - 0.5)
        return (torch.sigmoid(Lp) - torch.sigmoid(Lm)).clamp_min(1e-12)


class GaussianConditional(EntropyModel):
    """EntropyModels.py:188-207: p = Phi((x+.5-mu)/sigma) - Phi((x-.5-mu)/sigma)."""

    def likelihood_and_log(self, x: Tensor, mu: Tensor, sigma: Tensor, bound=None):
        params = torch.cat([mu.expand_as(x), sigma.expand_as(x)], dim=1)
        return F_.gmm_likelihood(x, params, 1, self.likelihood_lower_bound if bound is None else bound)

    def packed_likelihood_and_log(self, x: Tensor, params: Tensor, K: int):
        """params: the packed activation tensor produced by EntropyParameters.packed()."""
        return F_.gmm_likelihood(x, params, K, self.likelihood_lower_bound)

    def discretized_gaussian_pmf(self, x, mu, sigma):
        return self.likelihood_and_log(x, mu, sigma, bound=0.0)[0]

    def _likelihood(self, x: Tensor, mu: Tensor, sigma: Tensor) -> Tensor:
        return self.discretized_gaussian_pmf(x, mu, sigma)


class GaussianMixtureConditional(GaussianConditional):
    """EntropyModels.py:210-233: p = sum_k w_k pmf_k(x); weights/mus/sigmas are [B,K,M,h,w]."""

    def likelihood_and_log(self, x: Tensor, weights: Tensor, mus: Tensor, sigmas: Tensor, bound=None):
        B, K, M = weights.shape[:3]
        params = torch.cat([t.reshape(B, K * M, *t.shape[3:]) for t in (weights, mus, sigmas)], dim=1)
        return F_.gmm_likelihood(x, params, K, self.likelihood_lower_bound if bound is None else bound)

    def discretized_mixture_pmf(self, x, weights, mus, sigmas):
        return self.likelihood_and_log(x, weights, mus, sigmas, bound=0.0)[0]

    def _likelihood(self, x, weights, mus, sigmas) -> Tensor:
        return self.discretized_mixture_pmf(x, weights, mus, sigmas)


class MaskedConv2d(Conv2d):
    """PixelCNN masked convolution (ContextModels.py:3-20): the weight PARAMETER is zeroed in
    place on every forward; its gradient is not masked."""

    def __init__(self, mask_type, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert mask_type in ('A', 'B')
        self.register_buffer('mask', self.weight.data.clone())
        _, _, kH, kW = self.weight.size()
        self.mask.fill_(1)
        self.mask[:, :, kH // 2, kW // 2 + (mask_type == 'B'):] = 0
        self.mask[:, :, kH // 2 + 1:] = 0
        live = 0
        for r in range(kH):
            for s in range(kW):
                if r < kH // 2 or (r == kH // 2 and s < kW // 2 + (mask_type == 'B')):
                    live |= 1 << (r * kW + s)
        self._tap_mask = live

    def forward(self, x: Tensor, bf16: bool = False, out=None) -> Tensor:
        if F_.prepared(self.weight, "masked") is None:  # else this step's lic_prep_run already masked it in place
            F_.mask_weight_(self.weight, self.mask)
        s, p = self.stride[0], self.padding[0]
        if bf16:  # bf16 features out (they feed the bf16 entropy-parameter MLP)
            return FB_.conv2d_bf16(x, self.weight, self.bias, s, p, False, False, 0.01, self._tap_mask, out)
        return F_.conv2d(x, self.weight, self.bias, s, p, False, 0.01, self._tap_mask, None, out)


class ContextModel(nn.Module):
    precision = "fp32"  # "bf16": bf16 operands / features (models.set_precision)

    def __init__(self, latent_channels=192):
        super().__init__()
        self.masked = MaskedConv2d("A", in_channels=latent_channels, out_channels=2 * latent_channels,
                                   kernel_size=5, stride=1, padding=2)

    def forward(self, x, out=None):
        return self.masked(x, bf16=self.precision == "bf16", out=out)


class EntropyParameters(nn.Module):
    """1x1-conv MLP 4M -> 640 -> 640 -> {2M | 3KM} (ParametersModels.py:8-64)."""
    precision = "fp32"

    def __init__(self, latent_channels=192, hyper_latent_channels=192, K=1):
        super().__init__()
        if not isinstance(K, int) or K < 1:
            raise ValueError(f"K must be int >= 1, got {K}")
        self.K = K
        self.distribution = 'Mean-Scale Gaussian' if K == 1 else 'Mixture of Gaussians'
        self.latent_channels = latent_channels
        self.hyper_latent_channels = hyper_latent_channels
        cin = 2 * latent_channels + 2 * hyper_latent_channels
        cout = 2 * latent_channels if K == 1 else 3 * K * latent_channels
        self.net = nn.Sequential(Conv2d(cin, 640, kernel_size=1), LeakyReLU(),
                                 Conv2d(640, 640, kernel_size=1), LeakyReLU(),
                                 Conv2d(640, cout, kernel_size=1))

    def packed(self, combined_feat: Tensor) -> Tensor:
        """Activated parameters as ONE tensor [B, G*K*M, h, w] (what the likelihood kernel reads)."""
        if self.precision == "bf16":  # bf16 hidden features, fp32 raw parameters out of the last layer
            raw = run_bf16(self.net, combined_feat, out_f32=True)
        else:
            raw = run_fused(self.net, combined_feat)
        return F_.entropy_params_activation(raw, self.latent_channels, self.K)

    def split(self, act: Tensor):
        M, K = self.latent_channels, self.K
        if K == 1:
            return act[:, :M], act[:, M:]
        t = K * M
        return tuple(act[:, i * t:(i + 1) * t].unflatten(1, (K, M)) for i in range(3))

    def forward(self, combined_feat: Tensor):
        return self.split(self.packed(combined_feat))
