"""CPU restatement of the entropy-coder pieces -- TEST INFRASTRUCTURE ONLY.

The reference has no coder, so there is nothing upstream to pin against: the tables
restate the reference's own distributions (`channel_cdf`, EntropyModels.py:171-174; the Gaussian /
mixture CDF of EntropyModels.py:192-233 with utils.py:6-8) in numpy/torch, and the range coder is
restated in pure Python (small cases only) so the C++ coder's BYTES can be compared, not just its
round trip."""
from __future__ import annotations

import math

import numpy as np
import torch


def quantize_cdf(F: np.ndarray) -> np.ndarray:
    """F: [..., S+1] CDF at the symbol edges with F[...,0]=0, F[...,S]=1 -> uint32 cum tables
    cum[i] = floor(F_i (65536 - S)) + i, made non-decreasing first."""
    S = F.shape[-1] - 1
    f32 = np.clip(F.astype(np.float32), 0.0, 1.0)
    c = np.floor(f32 * np.float32(65536 - S)).astype(np.int64)
    c = np.minimum(c, 65536 - S)
    c = np.maximum.accumulate(c, axis=-1)
    c[..., S] = 65536 - S
    return (c + np.arange(S + 1)).astype(np.uint32)


def factorized_tables(fe_module_cpu, lo: int, S: int) -> np.ndarray:
    """fe_module_cpu: any object with channel_cdf(ch, xs) (the reference's module or a restatement)."""
    C = fe_module_cpu.channels
    F = np.zeros((C, S + 1), np.float32)
    xs = torch.arange(lo, lo + S + 1, dtype=torch.float32) - 0.5
    for c in range(C):
        F[c] = fe_module_cpu.channel_cdf(c, xs).numpy()
    F[:, 0], F[:, S] = 0.0, 1.0
    return quantize_cdf(F)


def gmm_tables(weights, mus, sigmas, W: int):
    """weights/mus/sigmas: float32 arrays [K, N] (K=1: weights all ones).  Returns (center [N] int32,
    tables [N][2W+2] uint32) with F(x) = sum_k w_k * 0.5 (1 + erf((x-mu_k)/(sigma_k sqrt 2)))."""
    w, mu, sg = (torch.from_numpy(np.asarray(a, np.float32)) for a in (weights, mus, sigmas))
    mean = torch.zeros(mu.shape[1], dtype=torch.float32)
    for k in range(mu.shape[0]):
        mean = mean + w[k] * mu[k]
    center = torch.round(mean).to(torch.int32)
    S = 2 * W + 1
    i = torch.arange(S + 1, dtype=torch.float32)
    x = (center.float()[:, None] - W + i[None, :]) - 0.5          # [N, S+1]
    F = torch.zeros_like(x)
    for k in range(mu.shape[0]):
        t = (x - mu[k][:, None]) / sg[k][:, None]
        F = F + w[k][:, None] * (0.5 * (1.0 + torch.erf(t / 1.41421356237309515)))
    F[:, 0], F[:, S] = 0.0, 1.0
    return center.numpy(), quantize_cdf(F.numpy())


# ---- pure-Python range coder (the algorithm of lic_rangecoder.cpp, for byte-level comparison) ----
class _Enc:
    def __init__(self):
        self.low, self.range, self.cache, self.cache_size, self.out = 0, 0xFFFFFFFF, 0, 1, bytearray()

    def _shift_low(self):
        if (self.low & 0xFFFFFFFF) < 0xFF000000 or (self.low >> 32) != 0:
            carry, temp = (self.low >> 32) & 0xFF, self.cache
            while True:
                self.out.append((temp + carry) & 0xFF)
                temp = 0xFF
                self.cache_size -= 1
                if self.cache_size == 0:
                    break
            self.cache = (self.low >> 24) & 0xFF
        self.cache_size += 1
        self.low = (self.low & 0x00FFFFFF) << 8

    def encode(self, lo, hi):
        r = self.range >> 16
        self.low += r * lo
        self.range = (r * (hi - lo)) & 0xFFFFFFFF
        while self.range < (1 << 24):
            self.range = (self.range << 8) & 0xFFFFFFFF
            self._shift_low()

    def bit(self, b):
        self.encode(32768 if b else 0, 65536 if b else 32768)

    def gamma(self, v):
        x = v + 1
        nb = x.bit_length() - 1
        for _ in range(nb):
            self.bit(0)
        self.bit(1)
        for i in range(nb - 1, -1, -1):
            self.bit((x >> i) & 1)

    def finish(self):
        for _ in range(5):
            self._shift_low()
        return bytes(self.out)


def rc_encode(tables: np.ndarray, idx, table_of=None) -> bytes:
    S = tables.shape[-1] - 1
    e = _Enc()
    for n, v in enumerate(idx):
        t = tables[table_of[n] if table_of is not None else n]
        s = 0 if v <= 0 else (S - 1 if v >= S - 1 else int(v))
        e.encode(int(t[s]), int(t[s + 1]))
        if s == 0:
            e.gamma(-int(v))
        if s == S - 1:
            e.gamma(int(v) - (S - 1))
    return e.finish()


def ideal_bits(tables: np.ndarray, idx, table_of=None) -> float:
    S = tables.shape[-1] - 1
    bits = 0.0
    for n, v in enumerate(idx):
        t = tables[table_of[n] if table_of is not None else n]
        s = 0 if v <= 0 else (S - 1 if v >= S - 1 else int(v))
        bits += 16.0 - math.log2(int(t[s + 1]) - int(t[s]))
        if s in (0, S - 1):
            x = (-int(v) if s == 0 else int(v) - (S - 1)) + 1
            bits += 2 * (x.bit_length() - 1) + 1
    return bits
