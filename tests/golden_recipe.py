"""Deterministic, numpy-only recipe for parameters / inputs used by the golden fixtures.

The golden generator (oracle/make_golden.py, run once in the build container where the
reference's Python modules are importable) loads these values INTO the reference modules;
the tests regenerate the same values from the same recipe on any machine, so fixtures only
need to store expected outputs.  No reference code here -- just random numbers with sensible
magnitudes per state-dict key.
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

PEDESTAL = float(2.0 ** -36)


def _rng(key: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def make_param(key: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    r = _rng(key, seed)
    shape = tuple(int(s) for s in shape)
    leaf = key.split(".")[-1]
    if leaf == "pedestal":
        return np.full(shape, PEDESTAL, np.float32)
    if leaf == "bound":
        minimum = 1e-6 if "beta_reparam" in key else 0.0
        return np.full(shape, (minimum + PEDESTAL) ** 0.5, np.float32)
    if leaf == "mask":  # type-A mask (5x5): rows above centre + left of centre
        m = np.ones(shape, np.float32)
        kH, kW = shape[2], shape[3]
        m[:, :, kH // 2, kW // 2:] = 0
        m[:, :, kH // 2 + 1:] = 0
        return m
    if leaf == "beta":
        v = np.sqrt(r.uniform(0.5, 1.5, shape) + PEDESTAL)
        low = r.rand(*shape) < 0.15  # exercise the LowerBound branch
        v[low] = 5e-4
        return v.astype(np.float32)
    if leaf == "gamma":
        C = shape[0]
        v = r.uniform(0.0, 0.12, shape)
        v[np.arange(C), np.arange(C)] = np.sqrt(r.uniform(0.05, 0.15, C) + PEDESTAL)
        low = r.rand(*shape) < 0.2
        low[np.arange(C), np.arange(C)] = False
        v[low] = 1e-6  # below bound 2**-18
        return v.astype(np.float32)
    if re.search(r"(^|\.)matrices\.\d+$", key):
        out = shape[1]
        init = np.log(np.expm1(1.0 / (10.0 ** 0.25) / out))
        return (init + 0.3 * r.randn(*shape)).astype(np.float32)
    if re.search(r"(^|\.)biases\.\d+$", key):
        return r.uniform(-0.5, 0.5, shape).astype(np.float32)
    if re.search(r"(^|\.)factors\.\d+$", key):
        return (0.5 * r.randn(*shape)).astype(np.float32)
    if leaf == "bias":
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)
    if leaf == "weight" and len(shape) == 4:
        fan = shape[1] * shape[2] * shape[3]
        a = 1.5 / np.sqrt(fan)
        return r.uniform(-a, a, shape).astype(np.float32)
    raise KeyError(f"no recipe for {key} {shape}")


def make_state(keys_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, np.ndarray]:
    return {k: make_param(k, s, seed) for k, s in keys_shapes}


def make_image(B: int, H: int, W: int, seed: int) -> np.ndarray:
    """ToTensor()-range input (Dataloader.py:7-9): smooth-ish random image in [0,1)."""
    r = np.random.RandomState(seed)
    base = r.rand(B, 3, H // 8 + 1, W // 8 + 1)
    up = np.kron(base, np.ones((1, 1, 8, 8)))[:, :, :H, :W]
    x = 0.7 * up + 0.3 * r.rand(B, 3, H, W)
    return np.clip(x, 0.0, 0.999).astype(np.float32)


def make_noise(shape, seed: int) -> np.ndarray:
    return np.random.RandomState(seed).rand(*shape).astype(np.float32)
