"""rd_loss with the reference's signature and result dict (RateDistortionLoss.py:5-49), computed by
one fused device reduction and ONE device-to-host copy (the reference syncs 8 times via .item())."""
from __future__ import annotations

import torch

from . import functional as F_

_KEYS = ("loss", "bpp_y", "bpp_z", "bpp_total", "mse", "psnr", "bits_y", "bits_z", "bits_total")


def rd_loss(model_out: dict, x: torch.Tensor, lambda_rd: float, sync: bool = True):
    """sync=False returns 0-d device tensors instead of Python floats (no host sync at all)."""
    B = x.size(0)
    buf = F_.rd_loss_buffer(model_out['logp_y'], model_out['logp_z'], model_out['x_hat'], x, lambda_rd)
    det = buf.detach()
    res = {'loss': buf[0]}
    if sync:
        host = det[:9].tolist()
        for i, k in enumerate(_KEYS[1:], start=1):
            res[k] = host[i]
    else:
        for i, k in enumerate(_KEYS[1:], start=1):
            res[k] = det[i]
    res['mse_per_image'] = det[16:16 + B]
    res['psnr_per_image'] = det[16 + B:16 + 2 * B]
    return res
