"""rd_loss with the reference's signature and result dict (RateDistortionLoss.py:5-49), computed by
one fused device reduction and ONE device-to-host copy (the reference syncs 8 times via .item())."""
from __future__ import annotations

import torch

from . import functional as F_

_KEYS = ("loss", "bpp_y", "bpp_z", "bpp_total", "mse", "psnr", "bits_y", "bits_z", "bits_total")


def rd_loss(model_out: dict, x: torch.Tensor, lambda_rd: float, sync: bool = True):
    """sync=False returns 0-d device tensors instead of Python floats (no host sync at all)."""
    B = x.size(0)
    buf, loss = F_.rd_loss_buffer(model_out['logp_y'], model_out['logp_z'], model_out['x_hat'], x, lambda_rd)
    det = buf.detach()
    res = {'loss': loss}
    if sync:
        host = det[:9].tolist()
        for i, k in enumerate(_KEYS[1:], start=1):
            res[k] = host[i]
    else:
        for i, k in enumerate(_KEYS[1:], start=1):
            res[k] = det[i]
        res['_buffer'] = det   # (all of the above in one tensor: `det[:9].tolist()` is ONE device-to-host copy)
    res['mse_per_image'] = det[16:16 + B]
    res['psnr_per_image'] = det[16 + B:16 + 2 * B]
    return res


def vision_rd_loss(model_out: dict, x: torch.Tensor, lambda_rd: float, gamma: float, frozen_activation=None, V=None):
    """RateDistortionLoss.py:52-121 for `ScalableImageCoding`: loss = bpp_y1 + bpp_y2 + bpp_z + lambda * mse (no 255^2
    factor here, :98), mse = reconstruction MSE (+ gamma * mean((frozen_activation(F_tilde) - V(x_hat))^2) when both
    modules are given).  Same result keys.  The reconstruction term and the y1 / z rates come from the fused
    rd_loss reduction (lambda folded as lambda / 255^2); the y2 rate and the optional vision term -- arbitrary user
    modules -- are torch reductions.  One device-to-host copy for all scalars."""
    import math
    B = x.size(0)
    num_pixels = x.size(2) * x.size(3)
    buf, loss0 = F_.rd_loss_buffer(model_out['logp_y1'], model_out['logp_z'], model_out['x_hat'], x, lambda_rd / 255.0 ** 2)
    bits_y2_img = -model_out['logp_y2'].sum(dim=(1, 2, 3)) / math.log(2.0)
    bpp_y2 = (bits_y2_img / num_pixels).mean()
    loss = loss0 + bpp_y2
    det = buf.detach()
    rec_mse_img, psnr_img = det[16:16 + B], det[16 + B:16 + 2 * B]
    vis = vis_img = None
    if frozen_activation is not None and V is not None:
        vis_img = ((frozen_activation(model_out['F_tilde']) - V(model_out['x_hat'])) ** 2).mean(dim=(1, 2, 3))
        vis = vis_img.mean()
        loss = loss + lambda_rd * gamma * vis
    # rd_loss_buffer layout: loss, bpp_y, bpp_z, bpp_total, mse, psnr, bits_y, bits_z, bits_total
    extra = torch.stack([bpp_y2.detach(), bits_y2_img.detach().mean(), vis.detach() if vis is not None else det[0] * 0])
    h = torch.cat([det[:9], extra]).tolist()
    bpp_y1, bpp_z, rec_mse, psnr, bits_y1, bits_z = h[1], h[2], h[4], h[5], h[6], h[7]
    bpp_y2_f, bits_y2_f, vis_f = h[9], h[10], h[11]
    has_v = vis is not None
    return {
        'loss': loss,
        'bpp_y1': bpp_y1, 'bpp_y2': bpp_y2_f, 'bpp_y': bpp_y1 + bpp_y2_f, 'bpp_z': bpp_z,
        'bpp_total': bpp_y1 + bpp_y2_f + bpp_z,
        'mse': rec_mse + (gamma * vis_f if has_v else 0.0), 'reconstruction_mse': rec_mse, 'psnr': psnr,
        'vision_mse': vis_f if has_v else 0.0,
        'mse_per_image': (rec_mse_img + gamma * vis_img.detach()) if has_v else rec_mse_img,
        'reconstruction_mse_per_image': rec_mse_img, 'psnr_per_image': psnr_img,
        'vision_mse_per_image': vis_img.detach() if has_v else 0.0,
        'bits_y1': bits_y1, 'bits_y2': bits_y2_f, 'bits_y': bits_y1 + bits_y2_f, 'bits_z': bits_z,
        'bits_total': bits_y1 + bits_y2_f + bits_z,
    }
