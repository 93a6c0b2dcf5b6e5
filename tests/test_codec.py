"""Entropy coder.  CPU: the C++ range coder against its pure-Python restatement
(byte-identical streams), round trips incl. escapes and degenerate tables, size vs ideal.  GPU: the
device-built tables against the numpy restatement (exact integers) and coded bpp vs estimated bpp
of a model."""
import numpy as np
import pytest
import torch

from oracle import codec_ref as CR


def _tables(r, T, S):
    """random valid tables: positive frequencies summing to 65536"""
    f = r.gamma(0.3, 1.0, size=(T, S)) + 1e-9
    f = f / f.sum(1, keepdims=True)
    F = np.concatenate([np.zeros((T, 1)), np.cumsum(f, 1)], 1)
    F[:, -1] = 1.0
    return CR.quantize_cdf(F)


@pytest.fixture(scope="module")
def codec():
    import __graft_entry__ as G  # builds liblic_codec.so (g++) if missing; no GPU needed for it
    G.build_codec()
    from neural_image_compression_amd import codec as CD
    return CD


def test_quantized_tables_are_valid():
    t = _tables(np.random.RandomState(0), 50, 33)
    assert (t[:, 0] == 0).all() and (t[:, -1] == 65536).all()
    assert (np.diff(t.astype(np.int64), axis=1) >= 1).all()
    # a degenerate CDF (all mass in one bin) still leaves every symbol codable
    F = np.zeros((1, 10)); F[0, 5:] = 1.0
    q = CR.quantize_cdf(F)
    assert (np.diff(q.astype(np.int64), axis=1) >= 1).all() and q[0, -1] == 65536


def test_rangecoder_matches_python_restatement_and_round_trips(codec):
    r = np.random.RandomState(1)
    S, T, n = 17, 7, 4000
    t = _tables(r, T, S)
    tof = r.randint(0, T, size=n).astype(np.int32)
    idx = r.randint(-3, S + 3, size=n).astype(np.int32)          # includes both escapes
    idx[::97] = r.randint(-100000, 100000, size=idx[::97].size)   # and large excesses
    data = codec.rc_encode(t, idx, tof)
    assert data == CR.rc_encode(t, idx, tof), "C++ and Python coders must emit identical bytes"
    back = codec.rc_decode(data, t, n, tof)
    assert (back == idx).all()
    with pytest.raises(codec.CodecError):
        codec.rc_decode(data[: len(data) // 2], t, n, tof)        # truncated stream is detected


def test_rangecoder_per_symbol_tables_and_size(codec):
    r = np.random.RandomState(2)
    S, n = 65, 20000
    t = _tables(r, n, S)                                          # one table per symbol (the y stream's shape)
    # draw symbols FROM the tables so that the ideal size is the entropy
    u = r.randint(0, 65536, size=n)
    idx = np.array([np.searchsorted(t[i], u[i], side="right") - 1 for i in range(n)], np.int32)
    data = codec.rc_encode(t, idx)
    assert (codec.rc_decode(data, t, n) == idx).all()
    ideal = codec.rc_ideal_bits(t, idx)
    assert abs(ideal - CR.ideal_bits(t, idx)) < 1e-6 * ideal
    assert 8 * len(data) <= ideal * 1.001 + 64, (8 * len(data), ideal)   # coder overhead < 0.1 % + flush


def test_empty_and_single_symbol_streams(codec):
    t = _tables(np.random.RandomState(3), 1, 5)
    assert codec.rc_decode(codec.rc_encode(t, np.zeros(0, np.int32), np.zeros(0, np.int32)), t, 0,
                           np.zeros(0, np.int32)).size == 0
    one = codec.rc_encode(t, np.array([2], np.int32), np.array([0], np.int32))
    assert codec.rc_decode(one, t, 1, np.array([0], np.int32))[0] == 2


# ---- device tables + end to end -------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 3])
def test_device_tables_match_restatement(codec, K):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    import golden_recipe as R
    M = 16
    model = nic.JointAutoregressiveHierarchical(M, K)
    st = R.make_state([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 31)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.cuda()
    # z tables: the device's channel CDF vs the same module evaluated through its own channel_cdf on
    # the GPU (the restatement of EntropyModels.py:171-174 used by the golden tests)
    lo, S = -20, 41
    got = codec.factorized_tables(model.factorized_entropy_model, lo, S).cpu().numpy().view(np.uint32)

    class _Dev:  # channel_cdf through the parity-tested device path, returned on the host
        channels = M

        @staticmethod
        def channel_cdf(c, xs):
            return model.factorized_entropy_model.channel_cdf(c, xs.cuda()).cpu()
    ref = CR.factorized_tables(_Dev, lo, S)
    assert (np.abs(got.astype(np.int64) - ref.astype(np.int64)) <= 1).all()   # floor() of an fp32 product: +-1 count
    assert (got[:, 0] == 0).all() and (got[:, -1] == 65536).all() and (np.diff(got.astype(np.int64), axis=1) >= 1).all()
    # y tables
    r = np.random.RandomState(5)
    P, W = 37, 12
    G = 2 if K == 1 else 3
    raw = torch.from_numpy((r.randn(1, G * K * M, P, 1) * 2).astype(np.float32)).cuda().contiguous(memory_format=torch.channels_last)
    from neural_image_compression_amd import functional as F_
    act = F_.entropy_params_activation(raw, M, K)
    center, tabs = codec.gmm_tables(act, M, K, W)
    a = act.permute(0, 2, 3, 1).reshape(P, G * K * M).cpu().numpy()
    if K == 1:
        w_, mu_, sg_ = np.ones((1, P * M), np.float32), a[:, :M].reshape(1, -1), a[:, M:].reshape(1, -1)
    else:
        T = K * M
        w_ = a[:, :T].reshape(P, K, M).transpose(1, 0, 2).reshape(K, -1)
        mu_ = a[:, T:2 * T].reshape(P, K, M).transpose(1, 0, 2).reshape(K, -1)
        sg_ = a[:, 2 * T:].reshape(P, K, M).transpose(1, 0, 2).reshape(K, -1)
    c_ref, t_ref = CR.gmm_tables(w_, mu_, sg_, W)
    tabs = tabs.cpu().numpy().view(np.uint32)
    same_center = center.cpu().numpy().ravel() == c_ref
    assert same_center.mean() > 0.99                                 # rint of an fp32 sum: ties may differ
    d = np.abs(tabs[same_center].astype(np.int64) - t_ref[same_center].astype(np.int64))
    assert d.max() <= 2, d.max()                                      # erf + floor in fp32
    assert (np.diff(tabs.astype(np.int64), axis=1) >= 1).all() and (tabs[:, -1] == 65536).all()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 3])
def test_coded_bpp_matches_estimate_and_round_trips(codec, K):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    import golden_recipe as R
    M = 32
    model = nic.JointAutoregressiveHierarchical(M, K)
    st = R.make_state([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 41)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.cuda().eval()
    x = torch.from_numpy(R.make_image(2, 128, 192, 42)).cuda().contiguous(memory_format=torch.channels_last)
    cd = codec.LatentCodec(model, z_lo=-32, z_S=65, y_W=24)
    res = cd.compress(x)
    # round trips
    z_back = cd.decompress_z(res["strings"]["z"], res["z_shape"])
    assert torch.equal(z_back, res["z_in"])
    y_back = cd.decode_y_with_tables(res["strings"]["y"], res["_y_tables"], res["_y_center"], cd.y_W, res["shape"])
    assert torch.equal(y_back.cuda(), res["y_in"])
    # the coder is within 0.1 % (+ flush bytes) of the ideal size for its 16-bit tables ...
    npix = x.shape[0] * x.shape[2] * x.shape[3]
    for s in ("y", "z"):
        assert res[f"bpp_coded_{s}"] <= res[f"bpp_ideal_{s}"] * 1.001 + 64.0 / npix
    # ... and the tables cost little against the model's own estimate (the estimate clamps
    # likelihoods at 1e-9 and is not quantised to 16 bits, so small deviations both ways are expected)
    tot_c = res["bpp_coded_y"] + res["bpp_coded_z"]
    tot_e = res["bpp_est_y"] + res["bpp_est_z"]
    assert abs(tot_c - tot_e) <= 0.02 * tot_e + 128.0 / npix, (tot_c, tot_e)


@pytest.mark.gpu
@pytest.mark.parametrize("K,B,H,W,kind,M", [(1, 2, 64, 128, "jah", 32), (3, 1, 128, 64, "jah", 32), (3, 2, 64, 192, "hmr", 32),
                                            (1, 3, 128, 128, "jah", 64), (3, 1, 192, 64, "jah", 64), (3, 4, 64, 256, "jah", 64)])
def test_context_codec_full_round_trip(codec, K, B, H, W, kind, M):
    """compress -> bytes -> decompress through the masked-conv context (wavefront schedule): the decoded
    latents equal the encoder's exactly, x_hat equals the model's eval output, coded ~ estimated."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    import golden_recipe as R
    # M = 64: every per-pixel layer is 64-column tileable, so the coder pins ONE igemm tile (conv2d_prepacked
    # pin_tile) while the encoder (B*h*w pixels at once) and the decoder (a wavefront of <= B*h pixels per step)
    # would otherwise pick tiles by their very different batch sizes
    model = (nic.JointAutoregressiveHierarchical if kind == "jah" else nic.HierarchicalMixtureResidual)(M, K)
    st = R.make_state([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 51)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.cuda().eval()
    x = torch.from_numpy(R.make_image(B, H, W, 52)).cuda().contiguous(memory_format=torch.channels_last)
    cc = codec.ContextCodec(model, z_lo=-32, z_S=65, y_W=24)
    enc = cc.compress(x)
    assert len(enc["strings"]["y"]) == B
    dec = cc.decompress(enc["strings"], enc["shape"], enc["z_shape"])
    assert torch.equal(dec["z_hat"], enc["z_in"])
    assert torch.equal(dec["y_hat"], enc["y_in"]), "decoder tables diverged from the encoder's"
    with torch.no_grad():
        ref = model(x, training=False)
    assert torch.equal(dec["x_hat"], ref["x_hat"])
    npix = B * H * W
    assert abs(enc["bpp_coded"] - enc["bpp_est"]) <= 0.02 * enc["bpp_est"] + (64.0 * (B + 1)) / npix
    # the stream carries a CRC-32 of each image's latent symbols: a mismatch is reported, not decoded silently
    assert len(enc["strings"]["y_crc32"]) == B
    bad = dict(enc["strings"], y_crc32=[c ^ 1 for c in enc["strings"]["y_crc32"]])
    with pytest.raises(codec.CodecError):
        cc.decompress(bad, enc["shape"], enc["z_shape"])


@pytest.mark.gpu
def test_evaluator_reports_coded_bpp(codec, tmp_path):
    """CompressionEvaluator.evaluate(coded=True): the bitstream size sits next to the estimated rates"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    from neural_image_compression_amd.evaluator import CompressionEvaluator
    import golden_recipe as R
    model = nic.JointAutoregressiveHierarchical(32, 3)
    st = R.make_state([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 61)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.cuda()
    # Kodak-like minimum for MS-SSIM (side > 160)
    batches = [torch.from_numpy(R.make_image(1, 192, 256, 62 + i)).contiguous(memory_format=torch.channels_last)
               for i in range(2)]
    ev = CompressionEvaluator(model, batches, torch.device("cuda:0"), 0.01, save_dir=str(tmp_path))
    m, _, _ = ev.evaluate(nic.rd_loss, coded=True)
    assert "BPP(coded)" in m and m["BPP(coded)"] > 0
    assert abs(m["BPP(coded)"] - m["BPP(total)"]) <= 0.02 * m["BPP(total)"] + 128.0 / (192 * 256)
    m2, _, _ = ev.evaluate(nic.rd_loss)
    assert "BPP(coded)" not in m2 and m2["BPP(total)"] == m["BPP(total)"]


@pytest.mark.parametrize("h,w", [(1, 1), (4, 4), (4, 8), (32, 48), (7, 3)])
def test_wavefront_schedule_is_causal_and_complete(codec, h, w):
    """every latent pixel appears exactly once, and each live tap of the type-A 5x5 mask belongs to an
    earlier step (or lies outside the image)"""
    cc = object.__new__(codec.ContextCodec)
    cc.pad = 2
    steps = cc._wavefront(h, w)
    assert len(steps) <= w + 3 * (h - 1)
    step_of = -np.ones((h, w), np.int64)
    for t, (ii, jj) in enumerate(steps):
        assert (np.diff(ii) > 0).all()
        assert (step_of[ii, jj] == -1).all()
        step_of[ii, jj] = t
    assert (step_of >= 0).all()
    taps = [(r, s) for r in range(5) for s in range(5) if r < 2 or (r == 2 and s < 2)]   # 12 live taps
    assert len(taps) == 12
    for i in range(h):
        for j in range(w):
            for r, s in taps:
                a, b = i + r - 2, j + s - 2
                if 0 <= a < h and 0 <= b < w:
                    assert step_of[a, b] < step_of[i, j]
