"""Data side of the harness.  The reference reads TextZoom LMDBs (dataset/dataset.py:94-133,257-270)
and yields (images_hr [B,C,32,128], images_lr [B,C,16,64], label_strs); lmdb/PIL/torchvision are not
in this image, so the loader here serves seeded synthetic batches of the same shapes and ranges
 -- the LMDB pipeline is row N3 of section 8f."""
from ..utils.synth import make_batch


class SyntheticTextZoom:
    """Iterable with DataLoader-like len(); batches are generated on the CPU generator."""

    def __init__(self, batch_size, iters, seed=1234, mask=False):
        self.batch_size, self.iters, self.seed, self.mask = batch_size, iters, seed, mask

    def __len__(self):
        return self.iters

    def __iter__(self):
        for i in range(self.iters):
            lr, hr, labels = make_batch(self.batch_size, self.seed + i, in_planes=4 if self.mask else 3)
            yield hr, lr, labels
