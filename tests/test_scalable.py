"""SURVEY 8(f).4, second half: `ScalableImageCoding` / `LatentSpaceTransform` / `vision_rd_loss`
(Models.py:208-338, Components.py:125-153, RateDistortionLoss.py:52-121).  The reference's versions cannot
execute, so nothing upstream can pin them: the HIP path is compared with the torch-op
restatement of the same repaired definitions (oracle/torch_ref.py) -- parity unpinned, stated."""
import numpy as np
import pytest
import torch

import golden_recipe as R
import neural_image_compression_amd as nic


def test_scalable_model_surface_cpu():
    with pytest.raises(ValueError):
        nic.ScalableImageCoding(0, 1, 1)
    with pytest.raises(ValueError):
        nic.ScalableImageCoding(32, 16, 0)
    with pytest.raises(ValueError):
        nic.ScalableImageCoding(32, 32, 1)
    m = nic.ScalableImageCoding(192, 128, K=3)
    assert (m.M, m.M1, m.M2, m.H, m.K) == (192, 128, 64, 192, 3) and m.distribution == 'Mixture of Gaussians'
    sd = m.state_dict()
    for key, shape in (("context_model_1.masked.weight", (256, 128, 5, 5)), ("context_model_2.masked.weight", (128, 64, 5, 5)),
                       ("entropy_parameters_1.net.0.weight", (640, 2 * 128 + 2 * 192, 1, 1)),
                       ("entropy_parameters_2.net.4.weight", (3 * 3 * 64, 640, 1, 1)),
                       ("LST.URB1.subpel_conv.deconv.weight", (128, 128, 3, 3)), ("LST.RB2.conv1.weight", (128, 128, 3, 3)),
                       ("LST.URB3.igdn.gamma", (128, 128)), ("LST.conv.weight", (128, 128, 3, 3))):
        assert tuple(sd[key].shape) == shape, key
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat"))
    from Models import ScalableImageCoding           # the compat shims export the reference's names
    from Components import LatentSpaceTransform
    from RateDistortionLoss import vision_rd_loss
    assert ScalableImageCoding is nic.ScalableImageCoding and vision_rd_loss is nic.vision_rd_loss
    assert LatentSpaceTransform(16, [2, 1, 1, 2]).conv.out_channels == 32


@pytest.mark.gpu
@pytest.mark.parametrize("K,with_vision", [(1, False), (3, True)])
def test_scalable_model_step_vs_torch_restatement(K, with_vision):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from oracle import torch_ref as TR
    dev = torch.device("cuda:0")
    M, M1, B, H, lam, gamma = 32, 20, 2, 64, 40.0, 0.5
    model = nic.ScalableImageCoding(M, M1, K)
    st = R.make_state([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 71)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(dev)
    x = R.make_image(B, H, H, 72)
    uz, uy = R.make_noise((B, M, 1, 1), 73), R.make_noise((B, M, 4, 4), 74)
    act = V = act_d = V_d = None
    if with_vision:   # stand-ins for Extra.py's frozen YOLO halves: any two fixed modules with matching outputs
        torch.manual_seed(5)
        act, V = torch.nn.Tanh(), torch.nn.Sequential(torch.nn.Conv2d(3, M1, 3, stride=8, padding=1), torch.nn.Tanh())
        for q in V.parameters():
            q.requires_grad_(False)
        import copy
        act_d, V_d = act, copy.deepcopy(V).to(dev)
    tx = torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last)
    out = model(tx, noise=(torch.from_numpy(uz).to(dev), torch.from_numpy(uy).to(dev)))
    assert set(out) >= {"x_hat", "y", "y_in", "y1", "y2", "z", "z_in", "p_z", "logp_z", "p_y1", "logp_y1", "p_y2", "logp_y2",
                        "F_tilde", "training"} | ({"mu1", "sigma1", "mu2", "sigma2"} if K == 1 else
                                                  {"weights1", "mus1", "sigmas1", "weights2", "mus2", "sigmas2"})
    assert tuple(out["F_tilde"].shape) == (B, M1, 8, 8) and tuple(out["y2"].shape) == (B, M - M1, 4, 4)
    res = nic.vision_rd_loss(out, tx, lam, gamma, act_d, V_d)
    res["loss"].backward()
    t_out, t_res, t_g = TR.step_scalable(st, x, M, M1, K, (uz, uy), lam, gamma, act, V)
    for k in ("y", "x_hat", "F_tilde", "logp_y1", "logp_y2", "logp_z"):
        a, b = out[k].detach().cpu().numpy().astype(np.float64), t_out[k].astype(np.float64)
        assert (np.abs(a - b) <= 1e-4 + 1e-4 * np.abs(b)).all(), (k, np.abs(a - b).max())
    for k in ("bpp_y1", "bpp_y2", "bpp_z", "bpp_total", "mse", "reconstruction_mse", "psnr", "bits_y1", "bits_z") + \
            (("vision_mse",) if with_vision else ()):
        assert abs(res[k] - t_res[k]) <= 1e-4 * abs(t_res[k]) + 1e-9, (k, res[k], t_res[k])
    assert abs(float(res["loss"]) - t_res["loss"]) <= 1e-4 * abs(t_res["loss"])
    assert res["bits_total"] == pytest.approx(res["bits_y1"] + res["bits_y2"] + res["bits_z"])
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        if name not in t_g:     # without the vision term nothing depends on the latent-space transform
            assert name.startswith("LST.") and not with_vision and p.grad is None, name
            continue
        ref = t_g[name]
        e = max(0.0, float(np.abs(p.grad.detach().cpu().numpy() - ref).max()) - 3e-7) / max(np.abs(ref).max(), 1e-12)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] <= 1e-3, worst


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "vision"])
def test_vision_rd_loss_golden_gpu(tag):
    """the HIP path (loss.vision_rd_loss -> lic_rd_loss_* with lambda / 255^2 folded in) against the fixture the
    REFERENCE's vision_rd_loss produced (tests/golden/op_vision_rd_loss.npz, oracle/make_golden.py): every result
    key and every gradient, with and without the vision term"""
    import os
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    d = torch.device("cuda:0")
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "op_vision_rd_loss.npz"))
    act = V = None
    if tag == "vision":
        V = torch.nn.Conv2d(3, 5, 3, stride=2, padding=1)
        with torch.no_grad():
            V.weight.copy_(torch.from_numpy(fx["v_weight"]))
            V.bias.copy_(torch.from_numpy(fx["v_bias"]))
        V, act = V.to(d), torch.nn.Tanh()
    ts = {k: torch.from_numpy(fx[k]).to(d).requires_grad_(True) for k in ("logp_y1", "logp_y2", "logp_z", "x_hat", "F_tilde")}
    res = nic.vision_rd_loss(ts, torch.from_numpy(fx["x"]).to(d), float(fx["lambda_rd"]), float(fx["gamma"]),
                             frozen_activation=act, V=V)
    res["loss"].backward()

    def close(a, b, what, rel=1e-4):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert np.abs(a - b).max() <= rel * max(np.abs(b).max(), 1e-30) + 1e-9, (what, np.abs(a - b).max(), np.abs(b).max())

    close(float(res["loss"]), float(fx[f"{tag}.loss"]), "loss")
    for k in ("bpp_y1", "bpp_y2", "bpp_y", "bpp_z", "bpp_total", "mse", "reconstruction_mse", "psnr", "vision_mse", "bits_y1",
              "bits_y2", "bits_y", "bits_z", "bits_total"):
        close(res[k], float(fx[f"{tag}.{k}"]), k)
    for k in ("mse_per_image", "reconstruction_mse_per_image", "psnr_per_image"):
        close(res[k].cpu().numpy(), fx[f"{tag}.{k}"], k)
    for k in ("logp_y1", "logp_y2", "logp_z", "x_hat"):
        close(ts[k].grad.cpu().numpy(), fx[f"{tag}.d{k}"], "d" + k)
    if tag == "vision":
        close(res["vision_mse_per_image"].cpu().numpy(), fx["vision.vision_mse_per_image"], "vision_mse_per_image")
        close(ts["F_tilde"].grad.cpu().numpy(), fx["vision.dF_tilde"], "dF_tilde")
        close(V.weight.grad.cpu().numpy(), fx["vision.dv_weight"], "dV.weight")
