#!/usr/bin/env python3
"""Tile / split sweep over the latent-side layers of config 2 (developer tool, GPU only): for every layer
(forward and data-gradient launch) time lic_igemm (+ its split-K finish) for each forced (BM, TN, split)
through the descriptor overrides, next to the automatic choice.  usage: python tools/sweep_latent.py [filter]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import _lib as L  # noqa: E402
from neural_image_compression_amd import functional as F_  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
M = int(os.environ.get("M", "192"))
REPS = int(os.environ.get("REPS", "20"))
CTX_MASK = 0
for r in range(5):
    for s in range(5):
        if r < 2 or (r == 2 and s < 2):
            CTX_MASK |= 1 << (r * 5 + s)

H = 16
LAYERS = [
    # name, Hi, Cin, Ho, Cout, k, stride, pad, transposed, tap_mask
    ("henc1 3x3", H, M, H, M, 3, 1, 1, False, 0),
    ("henc2 5x5s2", H, M, H // 2, M, 5, 2, 2, False, 0),
    ("henc3 5x5s2", H // 2, M, H // 4, M, 5, 2, 2, False, 0),
    ("hdec1 T5x5", H // 4, M, H // 2, M, 5, 2, 2, True, 0),
    ("hdec2 T5x5", H // 2, M, H, M * 3 // 2, 5, 2, 2, True, 0),
    ("hdec3 3x3", H, M * 3 // 2, H, 2 * M, 3, 1, 1, False, 0),
    ("ctx 5x5m", H, M, H, 2 * M, 5, 1, 2, False, CTX_MASK),
    ("ep1 1x1", H, 4 * M, H, 640, 1, 1, 0, False, 0),
    ("ep2 1x1", H, 640, H, 640, 1, 1, 0, False, 0),
    ("ep3 1x1", H, 640, H, 2 * M, 1, 1, 0, False, 0),
]


# the RGB-side dense GEMMs (column buffers [pixels][80]) at 128x128: rows = B*128*128 pixels, 1x1 "layers"
P128 = 128 * 128
IMAGE_GEMMS = [
    ("stem/head-dx K80 N192", 80, 192),
    ("head fwd K192 N80", 192, 80),
]


def time_launch(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


def main():
    only = sys.argv[1:] or None
    for name, Hi, Cin, Ho, Cout, k, stride, pad, tr, mask in LAYERS:
        for direction in ("fwd", "dgrad"):
            tag = f"{name} {direction}"
            if only and not any(o in tag for o in only):
                continue
            if direction == "fwd":
                g = dict(B=B, Hi=Hi, Wi=Hi, Cin=Cin, Ho=Ho, Wo=Ho, Cout=Cout, kh=k, kw=k, stride=stride, pad=pad,
                         transposed=tr, tap_mask=mask)
            else:  # data gradient: the transposed gather with the channel roles swapped
                g = dict(B=B, Hi=Ho, Wi=Ho, Cin=Cout, Ho=Hi, Wo=Hi, Cout=Cin, kh=k, kw=k, stride=stride, pad=pad,
                         transposed=not tr, tap_mask=mask)
            x = torch.randn(B, g["Hi"], g["Wi"], g["Cin"], device=dev)
            w = torch.randn(k * k, g["Cin"], g["Cout"], device=dev) * 0.05
            wp = F_._pack(w, k * k, g["Cin"], g["Cout"], g["Cin"] * g["Cout"], g["Cout"], 1)
            out = torch.empty(B, g["Ho"], g["Wo"], g["Cout"], device=dev)
            bias = torch.randn(g["Cout"], device=dev)
            taps = bin(mask).count("1") if mask else k * k
            flops = 2.0 * B * g["Ho"] * g["Wo"] * g["Cin"] * g["Cout"] * taps
            if g["transposed"] and stride == 2:
                flops /= 4
            res = []
            cands = [None] + [(bm, tn, sp) for bm in (64, 128) for tn in (1, 2, 3) for sp in (1, 2, 3, 4, 6, 8, 12)]
            for c in cands:
                F_.FORCE_IGEMM = c
                try:
                    ms = time_launch(lambda: F_._igemm(x, wp, out, bias=bias, epilogue=L.EPI_LEAKY, **g))
                except L.LicError:
                    continue
                finally:
                    F_.FORCE_IGEMM = None
                res.append((ms, c))
            auto = [r for r in res if r[1] is None][0][0]
            res.sort(key=lambda r: r[0])
            best = ", ".join(f"{c}: {ms * 1e3:.0f}us {flops / ms / 1e9:.0f}TF" for ms, c in res[:4])
            print(f"{tag:20s} auto {auto * 1e3:6.0f}us {flops / auto / 1e9:5.0f}TF | {best}", flush=True)


def image_gemms():
    only = sys.argv[1:] or None
    for name, Kd, Nd in IMAGE_GEMMS:
        if only and not any(o in name for o in only):
            continue
        P = B * P128
        x = torch.randn(P, Kd, device=dev)
        w = torch.randn(Kd, Nd, device=dev) * 0.05
        wp = F_._pack_dense(w)
        out = torch.empty(P, Nd, device=dev)
        g = dict(B=1, Hi=1, Wi=P, Cin=Kd, Ho=1, Wo=P, Cout=Nd, kh=1, kw=1, stride=1, pad=0, transposed=False)
        flops = 2.0 * P * Kd * Nd
        byts = 4.0 * P * (Kd + Nd)
        res = []
        for c in [None] + [(bm, tn, 0) for bm in (64, 128) for tn in (1, 2, 3)]:
            F_.FORCE_IGEMM = c
            try:
                ms = time_launch(lambda: F_._igemm(x, wp, out, **g))
            except L.LicError:
                continue
            finally:
                F_.FORCE_IGEMM = None
            res.append((ms, c))
        res.sort(key=lambda r: r[0])
        print(f"{name:24s} " + ", ".join(f"{c}: {ms * 1e3:.0f}us {flops / ms / 1e9:.0f}TF {byts / ms / 1e9:.2f}TB/s"
                                          for ms, c in res), flush=True)


if __name__ == "__main__":
    image_gemms()
    main()
