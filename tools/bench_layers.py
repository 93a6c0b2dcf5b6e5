#!/usr/bin/env python3
"""Per-layer MFMA kernel timing for the config-2 shapes (developer tool, GPU only).
Prints, per layer and per launch kind, the kernel variant, average ms and achieved TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_image_compression_amd import functional as F_  # noqa: E402
from neural_image_compression_amd.layers import GDN  # noqa: E402

dev = torch.device("cuda:0")
B, M = 32, 192
REPS = int(os.environ.get("REPS", "3"))


def nhwc(b, c, h, w, grad=True):
    t = torch.randn(b, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    return t.requires_grad_(grad)


def run(name, fn, *tensors):
    outs = []
    for it in range(REPS + 1):
        F_.PROFILE = [] if it > 0 else None
        y = fn(*tensors)
        g = torch.randn_like(y)
        y.backward(g)
        torch.cuda.synchronize()
        if it > 0:
            outs.append(F_.PROFILE)
        for t in tensors:
            t.grad = None
    F_.PROFILE = None
    n = len(outs[0])
    for i in range(n):
        nm, flops, ab = outs[0][i][0], outs[0][i][1], outs[0][i][2]
        ms = sum(o[i][3].elapsed_time(o[i][4]) for o in outs) / len(outs)
        print(f"{name:28s} #{i} {nm:34s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF/s  alg {ab / ms / 1e6:7.1f} GB/s")


def main():
    only = sys.argv[1:] if len(sys.argv) > 1 else None

    def want(n):
        return only is None or any(o in n for o in only)

    for (hi, tag) in ((128, "enc conv2 128->64"), (64, "enc conv3 64->32"), (32, "enc conv4 32->16")):
        if want(tag):
            w = torch.randn(M, M, 5, 5, device=dev, requires_grad=True)
            b = torch.randn(M, device=dev, requires_grad=True)
            run(tag, lambda x, w, b: F_.conv2d(x, w, b, 2, 2), nhwc(B, M, hi, hi), w, b)
    for (hi, tag) in ((16, "dec convT1 16->32"), (32, "dec convT2 32->64"), (64, "dec convT3 64->128")):
        if want(tag):
            w = torch.randn(M, M, 5, 5, device=dev, requires_grad=True)
            b = torch.randn(M, device=dev, requires_grad=True)
            run(tag, lambda x, w, b: F_.conv_transpose2d(x, w, b, 2, 2, 1), nhwc(B, M, hi, hi), w, b)
    for (hi, tag) in ((128, "gdn 128"), (64, "gdn 64"), (32, "gdn 32")):
        if want(tag):
            m = GDN(M).to(dev)
            run(tag, lambda x: m(x), nhwc(B, M, hi, hi))
    if want("probe"):
        # short-K conv-gather (9 taps) vs transposed with the same per-workgroup work
        w = torch.randn(M, M, 3, 3, device=dev, requires_grad=True)
        b = torch.randn(M, device=dev, requires_grad=True)
        run("probe conv3x3 s1 64x64", lambda x, w, b: F_.conv2d(x, w, b, 1, 1), nhwc(B, M, 64, 64), w, b)
        run("probe convT3x3 s1 64x64", lambda x, w, b: F_.conv_transpose2d(x, w, b, 1, 1, 0), nhwc(B, M, 64, 64), w, b)
        w1 = torch.randn(M, M, 1, 1, device=dev, requires_grad=True)
        run("probe conv1x1 128x128", lambda x, w, b: F_.conv2d(x, w, b, 1, 0), nhwc(B, M, 128, 128), w1, b)
    if want("stem"):
        w = torch.randn(M, 3, 5, 5, device=dev, requires_grad=True)
        b = torch.randn(M, device=dev, requires_grad=True)
        run("stem conv 3->192 256->128", lambda x, w, b: F_.image_conv2d(x, w, b, 2, 2), nhwc(B, 3, 256, 256, False), w, b)
    if want("head"):
        w = torch.randn(M, 3, 5, 5, device=dev, requires_grad=True)
        b = torch.randn(3, device=dev, requires_grad=True)
        run("head convT 192->3 128->256", lambda x, w, b: F_.image_conv_transpose2d(x, w, b, 2, 2, 1),
            nhwc(B, M, 128, 128), w, b)
    for (ci, co, tag) in ((768, 640, "ep 768->640"), (640, 640, "ep 640->640"), (640, 384, "ep 640->384")):
        if want(tag):
            w = torch.randn(co, ci, 1, 1, device=dev, requires_grad=True)
            b = torch.randn(co, device=dev, requires_grad=True)
            run(tag, lambda x, w, b: F_.conv2d(x, w, b, 1, 0, True), nhwc(B, ci, 16, 16), w, b)
    if want("ctx"):
        w = torch.randn(2 * M, M, 5, 5, device=dev, requires_grad=True)
        b = torch.randn(2 * M, device=dev, requires_grad=True)
        run("ctx masked 5x5 192->384", lambda x, w, b: F_.conv2d(x, w, b, 1, 2, False, 0.01, 0xFFF), nhwc(B, M, 16, 16), w, b)
    for (ci, co, k, s, hi, tag) in ((M, M, 3, 1, 16, "henc conv3x3 16"), (M, M, 5, 2, 16, "henc conv5 16->8"),
                                    (M, M, 5, 2, 8, "henc conv5 8->4")):
        if want(tag):
            w = torch.randn(co, ci, k, k, device=dev, requires_grad=True)
            b = torch.randn(co, device=dev, requires_grad=True)
            run(tag, lambda x, w, b: F_.conv2d(x, w, b, s, k // 2, True), nhwc(B, ci, hi, hi), w, b)
    for (ci, co, hi, tag) in ((M, M, 4, "hdec convT5 4->8"), (M, 288, 8, "hdec convT5 8->16")):
        if want(tag):
            w = torch.randn(ci, co, 5, 5, device=dev, requires_grad=True)
            b = torch.randn(co, device=dev, requires_grad=True)
            run(tag, lambda x, w, b: F_.conv_transpose2d(x, w, b, 2, 2, 1, True), nhwc(B, ci, hi, hi), w, b)
    if want("hdec conv3x3"):
        w = torch.randn(2 * M, 288, 3, 3, device=dev, requires_grad=True)
        b = torch.randn(2 * M, device=dev, requires_grad=True)
        run("hdec conv3x3 288->384", lambda x, w, b: F_.conv2d(x, w, b, 1, 1), nhwc(B, 288, 16, 16), w, b)


if __name__ == "__main__":
    main()
