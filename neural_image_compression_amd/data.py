"""Input pipeline at GPU speed.

The reference's `PreprocessedDataset` (Dataloader.py:11-27) decodes one JPEG/PNG per item with PIL
and converts it with `ToTensor()` on the host: fp32 CHW in [0,1].  At ~1100 images/s per MI355X
that decode is the bottleneck, so the preprocessed 256x256 crops (preprocess.py writes them once)
are stored as **uint8 NHWC shards** and a batch is: one gather from a memory-mapped file -> one
pinned H2D copy of uint8 (4x fewer PCIe bytes than fp32) -> `lic_u8_to_f32` on the device
(`float(v) / 255`, bit-identical to `ToTensor()`), already in the channels_last layout the conv
kernels read.

Shard format (little endian): magic b"LICSHRD1", uint32 N, H, W, C, then N*H*W*C bytes.
`ShardLoader` is a drop-in for the `DataLoader` the reference's Trainer / Evaluator iterate
(`for imgs in loader`, `len(loader)`, re-iterable); it yields [B,3,H,W] tensors on the device.
"""
from __future__ import annotations

import os
import struct
from typing import Iterable, Iterator, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from . import functional as F_

MAGIC = b"LICSHRD1"
_HEADER = struct.Struct("<8sIIII")


def write_shard(path: str, images: np.ndarray) -> None:
    """images: uint8 [N, H, W, C]"""
    a = np.ascontiguousarray(images)
    if a.dtype != np.uint8 or a.ndim != 4:
        raise ValueError(f"images must be uint8 [N,H,W,C], got {a.dtype} {a.shape}")
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, *a.shape))
        f.write(a.tobytes())


def shard_from_image_files(files: Sequence[str], path: str) -> int:
    """Decode image files (the sorted jpg/jpeg/png list of Dataloader.py:13-18, `.convert("RGB")`)
    once, offline, into a shard.  All images must share one size (preprocess.py crops to 256x256)."""
    from PIL import Image  # offline tool only
    arrs = [np.asarray(Image.open(f).convert("RGB"), np.uint8) for f in files]
    if not arrs:
        raise ValueError("no images")
    if any(a.shape != arrs[0].shape for a in arrs):
        raise ValueError("images differ in size; shards hold one size")
    write_shard(path, np.stack(arrs))
    return len(arrs)


class ShardDataset:
    """Memory-mapped uint8 [N,H,W,C] shard(s); `ds[i]` -> uint8 [H,W,C] view."""

    def __init__(self, paths):
        if isinstance(paths, (str, os.PathLike)):
            paths = [paths]
        self._maps, self._starts = [], [0]
        shape = None
        for p in paths:
            with open(p, "rb") as f:
                magic, n, h, w, c = _HEADER.unpack(f.read(_HEADER.size))
            if magic != MAGIC:
                raise ValueError(f"{p}: not a LIC shard")
            if shape is not None and (h, w, c) != shape:
                raise ValueError(f"{p}: image shape {(h, w, c)} differs from {shape}")
            shape = (h, w, c)
            self._maps.append(np.memmap(p, np.uint8, "r", offset=_HEADER.size, shape=(n, h, w, c)))
            self._starts.append(self._starts[-1] + n)
        if shape is None:
            raise ValueError("no shards")
        self.image_shape = shape

    def __len__(self) -> int:
        return self._starts[-1]

    def __getitem__(self, i: int) -> np.ndarray:
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        s = int(np.searchsorted(self._starts, i, side="right")) - 1
        return self._maps[s][i - self._starts[s]]

    def gather(self, idx: Iterable[int], out: Optional[np.ndarray] = None) -> np.ndarray:
        idx = list(idx)
        if out is None:
            out = np.empty((len(idx),) + self.image_shape, np.uint8)
        for k, i in enumerate(idx):
            out[k] = self[i]
        return out


def u8_to_f32(batch_u8: torch.Tensor) -> torch.Tensor:
    """uint8 [B,H,W,C] on the device -> fp32 [B,C,H,W] (channels_last memory), values v/255."""
    if batch_u8.dtype != torch.uint8 or batch_u8.dim() != 4:
        raise ValueError("expected a uint8 [B,H,W,C] tensor")
    if not batch_u8.is_cuda:
        raise L.LicError("lic_u8_to_f32 needs a CUDA tensor (there is no CPU fallback)")
    src = batch_u8.contiguous()
    out = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    L.check(L.load().lic_u8_to_f32(F_._ptr(src), F_._ptr(out), src.numel(), F_._stream()), "lic_u8_to_f32")
    return out.permute(0, 3, 1, 2)


class ShardLoader:
    """Batches of a ShardDataset on `device`.  `rank`/`world_size` take every world_size-th batch slot
    of the (shuffled) order, so data-parallel ranks see disjoint images (the reference has no DP)."""

    def __init__(self, dataset: ShardDataset, batch_size: int, device, shuffle: bool = False, seed: int = 0,
                 drop_last: bool = False, rank: int = 0, world_size: int = 1):
        self.ds, self.bs, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.seed, self.drop_last = shuffle, int(seed), drop_last
        self.rank, self.world = int(rank), int(world_size)
        self.epoch = 0
        n = len(dataset) // self.world
        self._n_batches = n // self.bs if drop_last else (n + self.bs - 1) // self.bs

    def __len__(self) -> int:
        return self._n_batches

    def order(self) -> np.ndarray:
        idx = np.arange(len(self.ds))
        if self.shuffle:
            np.random.RandomState(self.seed + self.epoch).shuffle(idx)
        per = len(idx) // self.world
        return idx[self.rank * per:(self.rank + 1) * per]

    def __iter__(self) -> Iterator[torch.Tensor]:
        idx = self.order()
        self.epoch += 1
        h, w, c = self.ds.image_shape
        for b in range(self._n_batches):
            sel = idx[b * self.bs:(b + 1) * self.bs]
            host = torch.empty((len(sel), h, w, c), dtype=torch.uint8).pin_memory() \
                if self.device.type == "cuda" else torch.empty((len(sel), h, w, c), dtype=torch.uint8)
            self.ds.gather(sel, host.numpy())
            yield u8_to_f32(host.to(self.device, non_blocking=True))


# ---- logging statistics without full-tensor D2H copies (Trainer.py:167-217) -------------------
def tensor_stats(t: torch.Tensor, nbins: int = 64, lo: Optional[float] = None, hi: Optional[float] = None) -> dict:
    """count / mean / std / min / max / NaN count and an `nbins` histogram of a device tensor, computed
    on the device; only 6 doubles + nbins counters cross PCIe.  [lo, hi] defaults to the data range
    (one extra tiny pass)."""
    F_._require_cuda(t)
    x = t.detach().float().contiguous().view(-1)
    lib = L.load()
    ws = torch.empty(lib.lic_tensor_stats_workspace_bytes() // 8, device=x.device, dtype=torch.float64)
    stats = torch.empty(6, device=x.device, dtype=torch.float64)
    hist = torch.empty(nbins, device=x.device, dtype=torch.int64)

    def run(a, b):
        L.check(lib.lic_tensor_stats(F_._ptr(x), x.numel(), nbins, float(a), float(b), F_._ptr(stats), F_._ptr(hist),
                                     F_._ptr(ws), ws.numel() * 8, F_._stream()), "lic_tensor_stats")
        return stats.cpu().numpy(), hist.cpu().numpy()

    if lo is None or hi is None:
        s, _ = run(0.0, 1.0)
        lo_, hi_ = (s[3], s[4]) if s[0] > 0 else (0.0, 1.0)
        if not hi_ > lo_:
            hi_ = lo_ + 1.0
        lo, hi = (lo_ if lo is None else lo), (hi_ if hi is None else hi)
    s, h = run(lo, hi)
    n = max(s[0], 1.0)
    mean = s[1] / n
    var = max(s[2] / n - mean * mean, 0.0)
    return {"count": int(s[0]), "mean": float(mean), "std": float(var ** 0.5), "min": float(s[3]), "max": float(s[4]),
            "nan": int(s[5]), "lo": float(lo), "hi": float(hi), "hist": h.tolist()}
