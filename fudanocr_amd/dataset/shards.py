"""Pre-decoded TextZoom shards + the device-side half of `alignCollate_real`.

At 7 000 images/s per GPU (56 000 on a node) JPEG decoding and PIL resizing in DataLoader workers
(reference dataset/dataset.py:41-47,136-152,257-270, interfaces/base.py:91-136) cannot keep up, so the dataset is
converted ONCE (tools/textzoom_to_shards.py, or `write_shard` below) into flat uint8 arrays that already went through
the reference's `img.resize(size, Image.BICUBIC)`:

    <dir>/meta.json   {"n", "hr": [32,128,3], "lr": [16,64,3], "format": 1}
    <dir>/hr.u8       n x 32 x 128 x 3 bytes (NHWC)       <dir>/lr.u8   n x 16 x 64 x 3 bytes
    <dir>/labels.txt  n lines, the raw words (str_filt is applied at load time with the run's voc_type)

`ShardLoader` then does per batch: gather rows of the memory-mapped arrays into a PINNED staging buffer, one
asynchronous H2D copy on a copy stream (double-buffered: the next batch is in flight during the step), and ONE HIP
kernel per tensor that finishes the reference transform on the device -- ToTensor (/255, HWC -> CHW) and, with
`mask=True`, the mean-threshold mask channel of `resizeNormalize` (PIL's integer luma, threshold = the image's mean
luma, 255 where luma <= mean).  The result is bit-identical to the reference's CPU collate (tests/test_dataset.py).
"""
import ctypes
import json
import os

import numpy as np
import torch
from PIL import Image

from ..utils.util import str_filt

FORMAT = 1


def write_shard(path, samples, hr_size=(128, 32), lr_size=(64, 16)):
    """samples: iterable of (img_HR: PIL.Image, img_lr: PIL.Image, word: str) -- e.g. lmdbDataset_real(..., voc_type
    'all') items before filtering.  Applies the reference's bicubic resize (resizeNormalize.__call__, first line)."""
    os.makedirs(path, exist_ok=True)
    n = 0
    with open(os.path.join(path, "hr.u8"), "wb") as fh, open(os.path.join(path, "lr.u8"), "wb") as fl, \
            open(os.path.join(path, "labels.txt"), "w", encoding="utf-8") as ft:
        for img_hr, img_lr, word in samples:
            if "\n" in word:
                raise ValueError("label with a newline")
            fh.write(np.asarray(img_hr.convert("RGB").resize(hr_size, Image.BICUBIC), dtype=np.uint8).tobytes())
            fl.write(np.asarray(img_lr.convert("RGB").resize(lr_size, Image.BICUBIC), dtype=np.uint8).tobytes())
            ft.write(word + "\n")
            n += 1
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"n": n, "hr": [hr_size[1], hr_size[0], 3], "lr": [lr_size[1], lr_size[0], 3], "format": FORMAT}, f)
    return n


def is_shard(path):
    return os.path.isfile(os.path.join(path, "meta.json"))


class ShardDataset:
    """memory-mapped shard; len() and raw (hr u8, lr u8, word) access"""

    def __init__(self, path, voc_type="all", max_len=100):
        meta = json.load(open(os.path.join(path, "meta.json")))
        if meta.get("format") != FORMAT:
            raise ValueError("unknown shard format in %s" % path)
        self.n, self.hr_shape, self.lr_shape = meta["n"], tuple(meta["hr"]), tuple(meta["lr"])
        self.hr = np.memmap(os.path.join(path, "hr.u8"), dtype=np.uint8, mode="r", shape=(self.n,) + self.hr_shape)
        self.lr = np.memmap(os.path.join(path, "lr.u8"), dtype=np.uint8, mode="r", shape=(self.n,) + self.lr_shape)
        words = open(os.path.join(path, "labels.txt"), encoding="utf-8").read().split("\n")[:self.n]
        self.labels = [str_filt(w, voc_type) for w in words]
        self.voc_type, self.max_len = voc_type, max_len

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.hr[i], self.lr[i], self.labels[i]


def u8_to_input(u8_nhwc, mask=False):
    """device half of resizeNormalize: uint8 [B,H,W,3] (CUDA) -> float32 [B,3(+1),H,W] in [0,1] (HIP kernel)"""
    from .. import _lib
    if not (u8_nhwc.is_cuda and u8_nhwc.dtype == torch.uint8 and u8_nhwc.dim() == 4 and u8_nhwc.shape[3] == 3
            and u8_nhwc.is_contiguous()):
        raise RuntimeError("u8_to_input needs a contiguous uint8 CUDA tensor [B,H,W,3]")
    b, h, w, _ = u8_nhwc.shape
    out = torch.empty((b, 4 if mask else 3, h, w), device=u8_nhwc.device, dtype=torch.float32)
    _lib.call("focr_u8_to_input", ctypes.c_void_p(u8_nhwc.data_ptr()), ctypes.c_void_p(out.data_ptr()), b, h, w,
              int(bool(mask)), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out


class ShardLoader:
    """DataLoader-like iterable over one or several shards: yields (images_hr, images_lr, label_strs) with the image
    tensors ALREADY on `device` (pinned staging + asynchronous H2D on a copy stream, `prefetch` batches ahead).
    rank / world: every data-parallel process draws a disjoint slice of each epoch's permutation (same seed)."""

    def __init__(self, datasets, batch_size, device, shuffle=True, drop_last=True, mask=False, seed=1234, rank=0,
                 world=1, prefetch=2):
        self.sets = list(datasets) if isinstance(datasets, (list, tuple)) else [datasets]
        self.batch_size, self.device, self.shuffle, self.drop_last = batch_size, torch.device(device), shuffle, drop_last
        self.mask, self.seed, self.rank, self.world, self.prefetch = mask, seed, rank, world, max(1, prefetch)
        self.offsets = np.cumsum([0] + [len(d) for d in self.sets])
        self.epoch = 0
        self.copy_stream = torch.cuda.Stream(device=self.device)
        hs, ls = self.sets[0].hr_shape, self.sets[0].lr_shape
        self._stage = [(torch.empty((batch_size,) + hs, dtype=torch.uint8).pin_memory(),
                        torch.empty((batch_size,) + ls, dtype=torch.uint8).pin_memory())
                       for _ in range(self.prefetch + 1)]
        self._slot_event = [None] * len(self._stage)          # last H2D copy out of each pinned slot
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="focr-shard-gather")

    def _indices(self):
        n = int(self.offsets[-1])
        idx = np.random.RandomState(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        if self.world > 1:          # every rank must run the SAME number of steps (the gradient all-reduce is collective)
            idx = idx[:(n // self.world) * self.world][self.rank::self.world]
        return idx

    def __len__(self):
        n = int(self.offsets[-1]) // self.world
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _gather(self, slot, ids):
        """worker thread: host gather of one batch into pinned slot `slot`.  The slot's previous upload is waited for
        ON THE HOST first (`Event.synchronize`): a device-side `wait_event` orders streams, not the host's memcpy into
        the pinned buffer, and the training step never synchronises the host."""
        ev = self._slot_event[slot]
        if ev is not None:
            ev.synchronize()
            self._slot_event[slot] = None
        hr_p, lr_p = self._stage[slot]
        hr_np, lr_np = hr_p.numpy(), lr_p.numpy()
        ids = np.asarray(ids, dtype=np.int64)
        shard = np.searchsorted(self.offsets, ids, side="right") - 1
        labels = [None] * len(ids)
        for s in np.unique(shard):
            pos = np.nonzero(shard == s)[0]
            d, k = self.sets[int(s)], ids[pos] - int(self.offsets[int(s)])
            order = np.argsort(k, kind="stable")                 # ascending reads of the memory-mapped arrays
            hr_np[pos[order]] = d.hr[k[order]]
            lr_np[pos[order]] = d.lr[k[order]]
            for j, kk in zip(pos, k):
                labels[int(j)] = d.labels[int(kk)]
        return tuple(labels)

    def _upload(self, slot, nb):
        """launch thread: asynchronous H2D of a gathered slot on the copy stream -> (hr_dev_u8, lr_dev_u8, event)"""
        hr_p, lr_p = self._stage[slot]
        with torch.cuda.stream(self.copy_stream):
            hr_d = hr_p[:nb].to(self.device, non_blocking=True)
            lr_d = lr_p[:nb].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._slot_event[slot] = ev
        return hr_d, lr_d, ev

    def __iter__(self):
        idx = self._indices()
        self.epoch += 1
        nb = len(self)
        batches = [idx[i * self.batch_size:(i + 1) * self.batch_size] for i in range(nb)]
        ns = len(self._stage)
        # pipeline: gather(i + prefetch) runs in the worker thread while batch i trains; upload(i + 1) is issued as soon
        # as batch i was handed to the device transform, so it overlaps step i
        gathers = {i: self._pool.submit(self._gather, i % ns, batches[i]) for i in range(min(self.prefetch, nb))}
        nxt_gather = len(gathers)
        uploads = {}

        def ensure_upload(i):
            if i < nb and i not in uploads:
                labels = gathers.pop(i).result()
                uploads[i] = self._upload(i % ns, len(batches[i])) + (labels,)
        ensure_upload(0)
        for i in range(nb):
            hr_d, lr_d, ev, labels = uploads.pop(i)
            torch.cuda.current_stream().wait_event(ev)
            hr = u8_to_input(hr_d, self.mask)
            lr = u8_to_input(lr_d, self.mask)
            hr_d.record_stream(torch.cuda.current_stream())
            lr_d.record_stream(torch.cuda.current_stream())
            if nxt_gather < nb:
                # its pinned slot was last uploaded `prefetch + 1` batches ago; _gather waits for that upload on the host
                gathers[nxt_gather] = self._pool.submit(self._gather, nxt_gather % ns, batches[nxt_gather])
                nxt_gather += 1
            ensure_upload(i + 1)
            yield hr, lr, labels
