"""TextZoom input pipeline with the reference's API (scene-text-telescope/dataset/dataset.py:27-317):
`lmdbDataset`, `lmdbDataset_real`, `lmdbDataset_mix`, `resizeNormalize`, `alignCollate_syn`, `alignCollate_real`,
`ConcatDataset`, `randomSequentialSampler` -- same names, constructor arguments, return tuples and image maths
(PIL bicubic resize -> ToTensor -> optional mean-threshold mask channel).

Differences, all on the plumbing side:
  * `lmdb` is imported lazily (it is not in this image); without it the LMDB classes raise with a pointer to the
    pre-decoded shard format (fudanocr_amd/dataset/shards.py, tools/textzoom_to_shards.py), which is what the
    data-parallel trainer reads at 10^4-10^5 images/s;
  * torchvision is not needed: `to_tensor` restates `transforms.ToTensor` for uint8 PIL images (HWC -> CHW, /255).
`SyntheticTextZoom` (seeded TextZoom-shaped batches) is what bench.py and the parity tests consume.
"""
import bisect
import io
import random
import warnings

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset, sampler

from ..utils.synth import make_batch
from ..utils.util import str_filt

random.seed(0)
scale = 0.90


def to_tensor(img):
    """torchvision.transforms.ToTensor for 8-bit PIL images: [H,W,C] uint8 -> [C,H,W] float32 in [0,1]"""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.array(a.transpose(2, 0, 1), copy=True)).to(torch.float32).div(255)


def rand_crop(im):
    w, h = im.size
    p1 = (random.uniform(0, w * (1 - scale)), random.uniform(0, h * (1 - scale)))
    p2 = (p1[0] + scale * w, p1[1] + scale * h)
    return im.crop(p1 + p2)


def central_crop(im):
    w, h = im.size
    p1 = (((1 - scale) * w / 2), (1 - scale) * h / 2)
    p2 = ((1 + scale) * w / 2, (1 + scale) * h / 2)
    return im.crop(p1 + p2)


def buf2PIL(txn, key, type="RGB"):
    imgbuf = txn.get(key)
    buf = io.BytesIO()
    buf.write(imgbuf)
    buf.seek(0)
    return Image.open(buf).convert(type)


def _open_env(root):
    try:
        import lmdb
    except ImportError as e:
        raise ImportError("reading a TextZoom LMDB (%s) needs the `lmdb` module, which this image does not have; "
                          "convert the dataset once with tools/textzoom_to_shards.py where lmdb is available and "
                          "point TRAIN.train_data_dir at the shard directory (fudanocr_amd/dataset/shards.py)"
                          % root) from e
    env = lmdb.open(root, max_readers=1, readonly=True, lock=False, readahead=False, meminit=False)
    if not env:
        raise IOError("cannot create lmdb from %s" % root)
    return env


class _LmdbBase(Dataset):
    def __init__(self, root=None, voc_type="upper", max_len=100, test=False, env=None):
        super().__init__()
        self.env = env if env is not None else _open_env(root)      # env: any object with begin() -> txn.get(key)
        with self.env.begin(write=False) as txn:
            self.nSamples = int(txn.get(b"num-samples"))
        self.voc_type, self.max_len, self.test = voc_type, max_len, test

    def __len__(self):
        return self.nSamples


class lmdbDataset(_LmdbBase):
    """(image, label): HR image only (reference dataset.py:50-91)"""

    def __init__(self, root=None, voc_type="upper", max_len=31, test=True, env=None):
        super().__init__(root, voc_type, max_len, test, env)

    def __getitem__(self, index):
        assert index <= len(self), "index range error"
        index += 1
        txn = self.env.begin(write=False)
        word = str(txn.get(b"label-%09d" % index).decode())
        try:
            img = buf2PIL(txn, b"image_hr-%09d" % index, "RGB")
        except TypeError:
            img = buf2PIL(txn, b"image-%09d" % index, "RGB")
        except IOError:
            return self[index + 1]
        return img, str_filt(word, self.voc_type)


class lmdbDataset_real(_LmdbBase):
    """(img_HR 128x32, img_lr 64x16, label) (reference dataset.py:94-133)"""

    def __getitem__(self, index):
        assert index <= len(self), "index range error"
        index += 1
        txn = self.env.begin(write=False)
        word = str(txn.get(b"label-%09d" % index).decode())
        try:
            img_HR = buf2PIL(txn, b"image_hr-%09d" % index, "RGB")
            img_lr = buf2PIL(txn, b"image_lr-%09d" % index, "RGB")
        except IOError:
            return self[index + 1]
        return img_HR, img_lr, str_filt(word, self.voc_type)


class lmdbDataset_mix(_LmdbBase):
    """real LR half of the time, HR-as-LR otherwise (reference dataset.py:155-206)"""

    def __getitem__(self, index):
        assert index <= len(self), "index range error"
        index += 1
        txn = self.env.begin(write=False)
        word = str(txn.get(b"label-%09d" % index).decode())
        if self.test:
            try:
                img_HR = buf2PIL(txn, b"image_hr-%09d" % index, "RGB")
                img_lr = buf2PIL(txn, b"image_lr-%09d" % index, "RGB")
            except Exception:                                   # noqa: BLE001  (the reference catches everything)
                img_HR = buf2PIL(txn, b"image-%09d" % index, "RGB")
                img_lr = img_HR
        else:
            img_HR = buf2PIL(txn, b"image_hr-%09d" % index, "RGB")
            img_lr = buf2PIL(txn, b"image_lr-%09d" % index, "RGB") if random.uniform(0, 1) < 0.5 else img_HR
        return img_HR, img_lr, str_filt(word, self.voc_type)


class resizeNormalize(object):
    """PIL resize -> ToTensor -> optional mask channel: 255 where the luma is <= its image mean (dataset.py:136-152)"""

    def __init__(self, size, mask=False, interpolation=Image.BICUBIC):
        self.size, self.interpolation, self.mask = size, interpolation, mask

    def __call__(self, img):
        img = img.resize(self.size, self.interpolation)
        img_tensor = to_tensor(img)
        if self.mask:
            mask = img.convert("L")
            thres = np.array(mask).mean()
            mask = mask.point(lambda x: 0 if x > thres else 255)
            img_tensor = torch.cat((img_tensor, to_tensor(mask)), 0)
        return img_tensor


class randomSequentialSampler(sampler.Sampler):
    def __init__(self, data_source, batch_size):
        self.num_samples, self.batch_size = len(data_source), batch_size

    def __iter__(self):
        n_batch, tail = len(self) // self.batch_size, len(self) % self.batch_size
        index = torch.LongTensor(len(self)).fill_(0)
        i = -1
        for i in range(n_batch):
            start = random.randint(0, len(self) - self.batch_size)
            index[i * self.batch_size:(i + 1) * self.batch_size] = start + torch.arange(0, self.batch_size)
        if tail:
            start = random.randint(0, len(self) - self.batch_size)
            index[(i + 1) * self.batch_size:] = start + torch.arange(0, tail)
        return iter(index)

    def __len__(self):
        return self.num_samples


class alignCollate_syn(object):
    def __init__(self, imgH=64, imgW=256, down_sample_scale=4, keep_ratio=False, min_ratio=1, mask=False):
        self.imgH, self.imgW, self.keep_ratio, self.min_ratio = imgH, imgW, keep_ratio, min_ratio
        self.down_sample_scale, self.mask = down_sample_scale, mask

    def __call__(self, batch):
        images, label_strs = zip(*batch)
        transform = resizeNormalize((self.imgW, self.imgH), self.mask)
        transform2 = resizeNormalize((self.imgW // self.down_sample_scale, self.imgH // self.down_sample_scale),
                                     self.mask)
        images_hr = torch.cat([transform(image).unsqueeze(0) for image in images], 0)
        images_lr = [image.resize((image.size[0] // self.down_sample_scale, image.size[1] // self.down_sample_scale),
                                  Image.BICUBIC) for image in images]
        images_lr = torch.cat([transform2(image).unsqueeze(0) for image in images_lr], 0)
        return images_hr, images_lr, label_strs


class alignCollate_real(alignCollate_syn):
    def __call__(self, batch):
        images_HR, images_lr, label_strs = zip(*batch)
        transform = resizeNormalize((self.imgW, self.imgH), self.mask)
        transform2 = resizeNormalize((self.imgW // self.down_sample_scale, self.imgH // self.down_sample_scale),
                                     self.mask)
        images_HR = torch.cat([transform(image).unsqueeze(0) for image in images_HR], 0)
        images_lr = torch.cat([transform2(image).unsqueeze(0) for image in images_lr], 0)
        return images_HR, images_lr, label_strs


class ConcatDataset(Dataset):
    """concatenation of datasets, indexable on the fly (reference dataset.py:273-313)"""

    @staticmethod
    def cumsum(sequence):
        r, s = [], 0
        for e in sequence:
            s += len(e)
            r.append(s)
        return r

    def __init__(self, datasets):
        super().__init__()
        assert len(datasets) > 0, "datasets should not be an empty iterable"
        self.datasets = list(datasets)
        self.cumulative_sizes = self.cumsum(self.datasets)

    def __len__(self):
        return self.cumulative_sizes[-1]

    def __getitem__(self, idx):
        dataset_idx = bisect.bisect_right(self.cumulative_sizes, idx)
        sample_idx = idx if dataset_idx == 0 else idx - self.cumulative_sizes[dataset_idx - 1]
        return self.datasets[dataset_idx][sample_idx]

    @property
    def cummulative_sizes(self):
        warnings.warn("cummulative_sizes attribute is renamed to cumulative_sizes", DeprecationWarning, stacklevel=2)
        return self.cumulative_sizes


class SyntheticTextZoom:
    """Iterable with DataLoader-like len(): seeded synthetic TextZoom-shaped batches."""

    def __init__(self, batch_size, iters, seed=1234, mask=False):
        self.batch_size, self.iters, self.seed, self.mask = batch_size, iters, seed, mask

    def __len__(self):
        return self.iters

    def __iter__(self):
        for i in range(self.iters):
            lr, hr, labels = make_batch(self.batch_size, self.seed + i, in_planes=4 if self.mask else 3)
            yield hr, lr, labels
