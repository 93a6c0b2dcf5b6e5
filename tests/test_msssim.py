"""MS-SSIM: the CPU restatement's own invariants (no GPU), and the device kernel
(`lic_msssim`) against it.  pytorch-msssim is third-party and absent: parity unpinned, both sides
follow the package's published 0.2.1 algorithm."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as TR


def _pair(B, C, H, W, seed, noise=0.05):
    r = np.random.RandomState(seed)
    # smooth-ish image: low-res noise upsampled + fine noise, in [0, 1]
    base = torch.from_numpy(r.rand(B, C, (H + 15) // 16, (W + 15) // 16).astype(np.float32))
    x = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    x = (x + 0.1 * torch.from_numpy(r.rand(B, C, H, W).astype(np.float32))).clamp(0, 1)
    y = (x + noise * torch.from_numpy(r.randn(B, C, H, W).astype(np.float32))).clamp(0, 1)
    return x, y


def test_restatement_invariants():
    x, y = _pair(1, 3, 192, 176, 0)
    assert abs(float(TR.ms_ssim(x, x, data_range=1.0)) - 1.0) < 1e-6
    v1 = float(TR.ms_ssim(x, y, data_range=1.0))
    x2, y2 = _pair(1, 3, 192, 176, 0, noise=0.15)
    v2 = float(TR.ms_ssim(x2, y2, data_range=1.0))
    assert 0.0 < v2 < v1 < 1.0
    # data_range scaling: (255 x, 255 y, 255) == (x, y, 1)
    assert abs(float(TR.ms_ssim(255 * x, 255 * y, data_range=255.0)) - v1) < 2e-5
    per = TR.ms_ssim(torch.cat([x, x2]), torch.cat([y, y2]), data_range=1.0, size_average=False)
    assert per.shape == (2,) and abs(float(per[0]) - v1) < 1e-6 and abs(float(per[1]) - v2) < 1e-6
    with pytest.raises(ValueError):
        TR.ms_ssim(x[..., :160, :], y[..., :160, :], data_range=1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,layout", [(1, 3, 512, 768, "nhwc"),    # a Kodak frame, as the evaluator feeds it
                                            (2, 3, 355, 401, "nchw"),    # odd sides: padded pooling at every scale
                                            (1, 1, 512, 768, "nchw"),    # luma
                                            (3, 2, 161, 190, "nhwc")])   # smallest legal side
def test_msssim_device_vs_restatement(B, C, H, W, layout):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from neural_image_compression_amd import functional as F_
    x, y = _pair(B, C, H, W, H + W)
    ref_all = TR.ms_ssim(x, y, data_range=1.0, size_average=False)
    dx, dy = x.cuda(), y.cuda()
    if layout == "nhwc":
        dx, dy = dx.contiguous(memory_format=torch.channels_last), dy.contiguous(memory_format=torch.channels_last)
    got_all = F_.ms_ssim(dx, dy, data_range=1.0, size_average=False).cpu()
    assert torch.allclose(got_all, ref_all, rtol=2e-5, atol=2e-6), (got_all, ref_all)
    got = float(F_.ms_ssim(dx, dy, data_range=1.0))
    assert abs(got - float(ref_all.mean())) <= 2e-5
    assert abs(float(F_.ms_ssim(dx, dx, data_range=1.0)) - 1.0) < 1e-5
    # run-to-run bitwise reproducible (fixed-order reductions)
    assert float(F_.ms_ssim(dx, dy, data_range=1.0)) == got


@pytest.mark.gpu
def test_msssim_device_errors():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from neural_image_compression_amd import functional as F_
    a = torch.rand(1, 3, 160, 300, device="cuda")
    with pytest.raises(ValueError):
        F_.ms_ssim(a, a, data_range=1.0)
    with pytest.raises(Exception):
        F_.ms_ssim(torch.rand(1, 3, 200, 200), torch.rand(1, 3, 200, 200), data_range=1.0)  # CPU tensors
