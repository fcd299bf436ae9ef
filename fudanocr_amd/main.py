#!/usr/bin/env python3
"""Entry point with the reference's command line (scene-text-telescope/main.py:6-40):
  python -m fudanocr_amd.main --arch tbsrn --batch_size 128 --STN --exp_name X [--test | --demo] [--resume ckpt]
Multi-GPU: launch with torch.distributed.run, one process per GPU (replaces yaml `ngpu` + DataParallel):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m fudanocr_amd.main ...
TextBase.__init__ binds the process to cuda:LOCAL_RANK and joins the RCCL process group when WORLD_SIZE > 1."""
import argparse
import os

import yaml

from .interfaces.super_resolution import TextSR
from .utils.util import AttrDict


def main(config, args):
    mission = TextSR(config, args)
    if args.test:                      # same dispatch order as the reference (main.py:8-15)
        return mission.test()
    elif args.demo:
        return mission.demo()
    return mission.train()


def parse(argv=None):
    p = argparse.ArgumentParser(description="")
    p.add_argument("--arch", default="tbsrn", choices=["tbsrn", "tsrn"])
    p.add_argument("--text_focus", action="store_true", help="train with the text-focus loss (loss/text_focus_loss.py) instead of MSE + CRNN-CTC")
    p.add_argument("--stroke_focus", action="store_true", help="text-gestalt's criterion (loss/stroke_focus_loss.py): MSE + stroke_lambda * L1 of the stroke-level recognizer's attention maps when --text_focus is given")
    p.add_argument("--stroke_lambda", type=float, default=50)
    p.add_argument("--standin_assets", action="store_true",
                   help="allow --text_focus / --stroke_focus to run WITHOUT the pretrained recognizer, confuse.pkl or "
                        "english_decomposition.txt (none ships with the reference) on name-keyed weights / unit weights / a "
                        "seeded stroke table: benchmarking and tests only; without the flag a missing file raises "
                        "FileNotFoundError as in the reference")
    p.add_argument("--exp_name", required=True, help="Type your experiment name")
    p.add_argument("--test", action="store_true", default=False)
    p.add_argument("--test_data_dir", type=str, default="")
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--rec", default="crnn", choices=["crnn"])
    p.add_argument("--STN", action="store_true", default=False)
    p.add_argument("--syn", action="store_true", default=False)
    p.add_argument("--mixed", action="store_true", default=False)
    p.add_argument("--mask", action="store_true", default=False)
    p.add_argument("--hd_u", type=int, default=32)
    p.add_argument("--srb", type=int, default=5)
    p.add_argument("--demo", action="store_true", default=False)
    p.add_argument("--demo_dir", type=str, default="./demo")
    return p.parse_args(argv)


if __name__ == "__main__":
    args = parse()
    cfg_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "super_resolution.yaml")
    config = AttrDict(yaml.load(open(cfg_path), Loader=yaml.Loader))
    main(config, args)
