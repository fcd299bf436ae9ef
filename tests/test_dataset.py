"""Row N3 -- TextZoom input pipeline.  CPU: the reference-API dataset / collate classes against the fixture produced by
the REFERENCE's dataset/dataset.py (tests/golden/collate.npz, tools/make_golden_collate.py), the LMDB key protocol
through a stand-in environment, the shard round trip and PIL's integer luma.  GPU: the device half of the transform and
the prefetching shard loader, bit-identical to the reference's CPU collate."""
import io
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from fudanocr_amd.dataset import dataset as D
from fudanocr_amd.dataset import shards as S


def _fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "collate.npz"))
    words = json.load(open(os.path.join(golden_dir, "collate_labels.json")))
    n = len(words["words"])
    batch = [(Image.fromarray(g["in_hr_%d" % i]), Image.fromarray(g["in_lr_%d" % i]), words["words"][i]) for i in range(n)]
    return g, words, batch


def test_collate_matches_reference_fixture(golden_dir):
    g, words, batch = _fixture(golden_dir)
    for mask in (False, True):
        hr, lr, labels = D.alignCollate_real(imgH=32, imgW=128, down_sample_scale=2, mask=mask)(batch)
        assert labels == tuple(words["words"])
        assert torch.equal(hr, torch.tensor(g["real_hr_mask%d" % int(mask)]).float().div(255))
        assert torch.equal(lr, torch.tensor(g["real_lr_mask%d" % int(mask)]).float().div(255))
    hr, lr, _ = D.alignCollate_syn(imgH=32, imgW=128, down_sample_scale=2, mask=True)([(b[0], b[2]) for b in batch])
    assert torch.equal(hr, torch.tensor(g["syn_hr"]).float().div(255))
    assert torch.equal(lr, torch.tensor(g["syn_lr"]).float().div(255))
    from fudanocr_amd.utils.util import str_filt
    for voc, want in words["filtered"].items():
        assert [str_filt(w, voc) for w in words["words"]] == want


class _FakeTxn:
    def __init__(self, kv):
        self.kv = kv

    def get(self, k):
        return self.kv.get(k)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeEnv:
    """stands in for lmdb.Environment with TextZoom's key protocol (dataset/dataset.py:94-133)"""

    def __init__(self, samples):
        self.kv = {b"num-samples": str(len(samples)).encode()}
        for i, (hr, lr, w) in enumerate(samples, 1):
            for key, im in ((b"image_hr-%09d" % i, hr), (b"image_lr-%09d" % i, lr)):
                buf = io.BytesIO()
                im.save(buf, format="PNG")
                self.kv[key] = buf.getvalue()
            self.kv[b"label-%09d" % i] = w.encode()

    def begin(self, write=False):
        return _FakeTxn(self.kv)


def test_lmdb_dataset_protocol_and_missing_module(golden_dir):
    _, words, batch = _fixture(golden_dir)
    ds = D.lmdbDataset_real(voc_type="lower", env=_FakeEnv(batch))
    assert len(ds) == len(batch)
    hr, lr, lab = ds[2]
    assert lab == words["filtered"]["lower"][2] and hr.size == batch[2][0].size and lr.size == batch[2][1].size
    assert np.array_equal(np.asarray(hr), np.asarray(batch[2][0]))
    mix = D.lmdbDataset_mix(voc_type="all", env=_FakeEnv(batch), test=True)
    assert mix[0][2] == words["filtered"]["all"][0]
    cat = D.ConcatDataset([ds, ds])
    assert len(cat) == 2 * len(ds) and cat[len(ds) + 1][2] == ds[1][2]
    loader = torch.utils.data.DataLoader(cat, batch_size=4, shuffle=False, drop_last=True,
                                         collate_fn=D.alignCollate_real(imgH=32, imgW=128, down_sample_scale=2))
    hrb, lrb, labs = next(iter(loader))
    assert hrb.shape == (4, 3, 32, 128) and lrb.shape == (4, 3, 16, 64) and len(labs) == 4
    try:
        import lmdb  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="textzoom_to_shards"):
            D.lmdbDataset_real(root="/nonexistent")


def test_shard_round_trip(tmp_path, golden_dir):
    _, words, batch = _fixture(golden_dir)
    n = S.write_shard(str(tmp_path / "sh"), batch)
    assert n == len(batch) and S.is_shard(str(tmp_path / "sh")) and not S.is_shard(str(tmp_path))
    ds = S.ShardDataset(str(tmp_path / "sh"), voc_type="lower")
    assert len(ds) == n and ds.labels == words["filtered"]["lower"]
    for i, (hr, lr, _) in enumerate(batch):
        assert np.array_equal(ds.hr[i], np.asarray(hr.resize((128, 32), Image.BICUBIC)))
        assert np.array_equal(ds.lr[i], np.asarray(lr.resize((64, 16), Image.BICUBIC)))


def test_integer_luma_is_pils():
    """the device kernel's luma (19595 R + 38470 G + 7471 B + 0x8000) >> 16 is PIL's convert('L')"""
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, (64, 97, 3)).astype(np.uint8)
    want = np.asarray(Image.fromarray(a).convert("L")).astype(np.int64)
    x = a.astype(np.int64)
    got = (x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_device_transform_and_shard_loader(tmp_path, golden_dir):
    g, words, batch = _fixture(golden_dir)
    # device half of resizeNormalize vs the reference fixture, bit for bit (mask channel included)
    hr_u8 = torch.tensor(np.stack([np.asarray(b[0].resize((128, 32), Image.BICUBIC)) for b in batch])).cuda()
    for mask in (False, True):
        got = S.u8_to_input(hr_u8, mask)
        assert torch.equal(got.cpu(), torch.tensor(g["real_hr_mask%d" % int(mask)]).float().div(255))
    with pytest.raises(RuntimeError):
        S.u8_to_input(hr_u8.cpu(), False)
    # loader: several epochs-worth of batches, prefetching, two shards, rank slicing
    big = [batch[i % len(batch)] for i in range(23)]
    S.write_shard(str(tmp_path / "a"), big[:11])
    S.write_shard(str(tmp_path / "b"), big[11:])
    sets = [S.ShardDataset(str(tmp_path / d), voc_type="all") for d in ("a", "b")]
    loader = S.ShardLoader(sets, 4, "cuda:0", shuffle=False, drop_last=False, mask=True)
    assert len(loader) == 6
    seen = 0
    coll = D.alignCollate_real(imgH=32, imgW=128, down_sample_scale=2, mask=True)
    for hr, lr, labels in loader:
        ref_hr, ref_lr, ref_lab = coll(big[seen:seen + len(labels)])
        from fudanocr_amd.utils.util import str_filt
        assert labels == tuple(str_filt(w, "all") for w in ref_lab)          # the shard filters at load time
        assert torch.equal(hr.cpu(), ref_hr) and torch.equal(lr.cpu(), ref_lr)
        seen += len(labels)
    assert seen == 23
    parts = []
    for r in range(2):
        ld = S.ShardLoader(sets, 4, "cuda:0", shuffle=True, drop_last=True, seed=5, rank=r, world=2)
        parts.append([lab for _, _, labs in ld for lab in labs])
        assert len(parts[-1]) == 8                       # 23 samples -> 11 per rank -> 2 full batches on BOTH ranks
    a0 = S.ShardLoader(sets, 4, "cuda:0", shuffle=True, drop_last=True, seed=5, rank=0, world=2)
    assert [lab for _, _, labs in a0 for lab in labs] == parts[0]          # same seed, same epoch -> same order
