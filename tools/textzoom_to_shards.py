#!/usr/bin/env python3
"""Convert a TextZoom LMDB directory (keys num-samples, label-%09d, image_hr-%09d, image_lr-%09d; reference
dataset/dataset.py:94-133) into the pre-decoded shard format of fudanocr_amd/dataset/shards.py.  Needs the `lmdb`
module (not in the build image: run it wherever the dataset lives).

  python tools/textzoom_to_shards.py /data/TextZoom/train1 /data/shards/train1 [--width 128 --height 32 --scale 2]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fudanocr_amd.dataset import dataset, shards   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lmdb_dir")
    ap.add_argument("out_dir")
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--height", type=int, default=32)
    ap.add_argument("--scale", type=int, default=2)
    ap.add_argument("--max-len", type=int, default=100)
    a = ap.parse_args()
    ds = dataset.lmdbDataset_real(a.lmdb_dir, voc_type="all", max_len=a.max_len)

    def items():
        txn_env = ds.env
        for i in range(len(ds)):
            with txn_env.begin(write=False) as txn:
                word = str(txn.get(b"label-%09d" % (i + 1)).decode())
            hr, lr, _ = ds[i]
            yield hr, lr, word            # the RAW word: str_filt is applied at load time with the run's voc_type

    n = shards.write_shard(a.out_dir, items(), hr_size=(a.width, a.height),
                           lr_size=(a.width // a.scale, a.height // a.scale))
    print("wrote %d samples to %s" % (n, a.out_dir))


if __name__ == "__main__":
    main()
