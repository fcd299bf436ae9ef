"""TextBase: the reference's mission base class (interfaces/base.py:33-325) restated for one process per
GPU on the HIP modules.  Same method names and call order; DataParallel (base.py:178-179) is replaced by
fudanocr_amd.engine.TrainStep (flat buffers + RCCL all-reduce); the checkpoint dict keeps the reference
schema (base.py:255-272) with un-prefixed state_dict keys."""
import logging
import os
import shutil

import torch

from ..dataset import dataset
from ..dataset import shards
from ..dataset.dataset import SyntheticTextZoom
from ..loss.ctc_focus_loss import CTCFocusLoss
from ..model import tbsrn, tsrn
from ..model.crnn import crnn
from ..utils import ssim_psnr, util
from ..utils.utils_crnn import strLabelConverter
from ..utils.weight_fill import fill_module_
from .. import kernels as K


class TextBase(object):
    def __init__(self, config, args):
        self.config, self.args = config, args
        self.scale_factor = config.TRAIN.down_sample_scale
        self.mask = bool(getattr(args, "mask", False))
        self.resume = args.resume if args.resume is not None else config.TRAIN.resume
        self.batch_size = args.batch_size if args.batch_size is not None else config.TRAIN.batch_size
        self.exp_name = args.exp_name
        self.voc_type = config.TRAIN.voc_type
        if not torch.cuda.is_available():
            raise RuntimeError("fudanocr_amd needs an MI355X: there is no CPU fallback (the CPU oracle is test-only)")
        # one process per GPU (replaces nn.DataParallel, reference base.py:178-179): bind this process to its device
        # BEFORE any kernel runs (kernels launch on torch's current stream of the current device) and join the RCCL
        # process group, so that engine.TrainStep sees world > 1 and all-reduces the gradients
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        self.device = torch.device("cuda", local)
        if self.world > 1 and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("FOCR_DIST_BACKEND", "nccl")           # "gloo": ranks sharing one GPU (tests)
            if backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=self.device)
            else:
                torch.distributed.init_process_group(backend)
        alphabet = "0123456789abcdefghijklmnopqrstuvwxyz"
        self.converter_crnn = strLabelConverter(alphabet)
        self.cal_psnr = ssim_psnr.calculate_psnr
        self.cal_ssim = ssim_psnr.SSIM()
        self.ckpt_path = os.path.join("checkpoint", self.exp_name)
        if not args.test and not getattr(args, "demo", False) and self.rank == 0:
            if os.path.exists(self.ckpt_path) and not self.resume:
                shutil.rmtree(self.ckpt_path)              # reference base.py:80-84
            os.makedirs(self.ckpt_path, exist_ok=True)
            logging.basicConfig(format="%(message)s", level=logging.INFO, force=True,
                                handlers=[logging.FileHandler(os.path.join(self.ckpt_path, "log.txt")),
                                          logging.StreamHandler()])
        elif self.rank == 0:
            logging.basicConfig(format="%(message)s", level=logging.INFO, force=True)
        else:                                               # ranks > 0: no directory wipes, no log files
            logging.basicConfig(format="%(message)s", level=logging.WARNING, force=True)
        self.logging = logging

    # ---- data ------------------------------------------------------------------------------
    # reference interfaces/base.py:67-69,91-136: `load_dataset` / `align_collate` pick the dataset and collate classes,
    # get_train_data concatenates TRAIN.train_data_dir, get_val_data / get_test_data build one loader per directory.
    # A directory holding a `meta.json` is a pre-decoded shard (dataset/shards.py: pinned staging, async H2D, device
    # transform); anything else is opened as a TextZoom LMDB through the reference-API classes (needs `lmdb`).
    # With no directories configured the harness runs on seeded synthetic TextZoom-shaped batches.
    def _dataset_classes(self):
        a = self.args
        if getattr(a, "syn", False):
            return dataset.lmdbDataset, dataset.alignCollate_syn
        if getattr(a, "mixed", False):
            return dataset.lmdbDataset_mix, dataset.alignCollate_real
        return dataset.lmdbDataset_real, dataset.alignCollate_real

    @property
    def load_dataset(self):
        return self._dataset_classes()[0]

    @property
    def align_collate(self):
        return self._dataset_classes()[1]

    def _loader(self, dirs, test):
        cfg = self.config.TRAIN
        if all(shards.is_shard(d) for d in dirs):
            sets = [shards.ShardDataset(d, voc_type=cfg.voc_type, max_len=cfg.max_len) for d in dirs]
            loader = shards.ShardLoader(sets, self.batch_size, self.device, shuffle=not test, drop_last=not test,
                                        mask=self.mask, seed=cfg.manualSeed, rank=0 if test else self.rank,
                                        world=1 if test else self.world)
            return sets, loader
        sets = [self.load_dataset(root=d, voc_type=cfg.voc_type, max_len=cfg.max_len, test=test) for d in dirs]
        ds = dataset.ConcatDataset(sets)
        sampler = None
        if self.world > 1 and not test:
            sampler = torch.utils.data.distributed.DistributedSampler(ds, self.world, self.rank, shuffle=True,
                                                                      seed=int(cfg.manualSeed), drop_last=True)
        loader = torch.utils.data.DataLoader(
            ds, batch_size=self.batch_size, shuffle=(not test and sampler is None), sampler=sampler,
            num_workers=int(cfg.workers), drop_last=not test,
            collate_fn=self.align_collate(imgH=cfg.height, imgW=cfg.width, down_sample_scale=cfg.down_sample_scale,
                                          mask=self.mask))
        return ds, loader

    def get_train_data(self):
        cfg = self.config.TRAIN
        if not isinstance(cfg.train_data_dir, list):
            raise TypeError("check trainRoot")
        if cfg.train_data_dir:
            return self._loader(list(cfg.train_data_dir), test=False)
        # every rank draws its own shard of the global minibatch (seed + rank), as DataParallel's scatter did
        ds = SyntheticTextZoom(self.batch_size, int(getattr(cfg, "iters_per_epoch", 20)),
                               cfg.manualSeed + 1000 * self.rank, self.mask)
        return ds, ds

    def get_test_data(self, dir_):
        return self._loader([dir_], test=True)

    def get_val_data(self):
        cfg = self.config.TRAIN
        assert isinstance(cfg.VAL.val_data_dir, list)
        if cfg.VAL.val_data_dir:
            pairs = [self.get_test_data(d) for d in cfg.VAL.val_data_dir]
            return [p[0] for p in pairs], [p[1] for p in pairs]
        ds = SyntheticTextZoom(self.batch_size, 2, 99, self.mask)
        return [ds], [ds]

    # ---- model / optimiser --------------------------------------------------------------------
    def generator_init(self):
        a = self.args
        common = dict(scale_factor=self.scale_factor, width=self.config.TRAIN.width, height=self.config.TRAIN.height,
                      STN=a.STN, mask=self.mask, srb_nums=a.srb, hidden_units=a.hd_u)
        if a.arch == "tbsrn":
            model = tbsrn.TBSRN(**common)
        elif a.arch == "tsrn":
            model = tsrn.TSRN(**common)
        else:
            raise ValueError("only the hot-path architectures are built: tbsrn, tsrn")
        model = model.to(self.device)
        if self.resume:
            self.logging.info("loading pre-trained model from %s " % self.resume)
            model.load_state_dict(torch.load(self.resume, map_location=self.device)["state_dict_G"])
        para_num = sum(p.numel() for p in model.parameters())
        self.logging.info("Total Parameters {}".format(para_num))
        rec, _ = self.CRNN_init() if getattr(a, "ctc", True) else (None, None)
        if getattr(a, "stroke_focus", False):
            # text-gestalt's criterion (text-gestalt/interfaces/base.py:162): MSE + stroke_lambda * L1 on the attention
            # maps of the stroke-level recognizer.  `--text_focus` switches its focus term on, as in the reference.
            from ..loss.stroke_focus_loss import StrokeFocusLoss
            crit = StrokeFocusLoss(a, device=self.device)
            self._note_standins(crit)
            return {"model": model, "crit": crit, "recognizer": rec}
        if getattr(a, "text_focus", False):
            # the reference's criterion for tbsrn / tsrn (interfaces/base.py:143-150): MSE + text-focus terms.  The
            # CRNN stays the eval-time recognizer only, exactly as in the reference.
            from ..loss.text_focus_loss import TextFocusLoss
            crit = TextFocusLoss(a, device=self.device)
            if getattr(a, "text_focus", False):
                crit.weight_table()                          # resolve confuse.pkl now: a missing file fails at start-up
            self._note_standins(crit)
            return {"model": model, "crit": crit, "recognizer": rec}
        return {"model": model, "crit": CTCFocusLoss(rec), "recognizer": rec}

    def _note_standins(self, crit):
        """stand-in assets (explicit --standin_assets runs only) go into the run's config and log, not just a warning"""
        used = list(getattr(crit, "standin_assets", []))
        self.config["standin_assets"] = used
        if used:
            self.logging.info("STAND-IN ASSETS in use (benchmark / test run, not a training run): %s" % ", ".join(used))

    def optimizer_init(self, model, crit):
        from ..engine import TrainStep
        cfg = self.config.TRAIN
        return TrainStep(model, crit, lr=cfg.lr, betas=(cfg.beta1, 0.999), max_norm=0.25)

    def CRNN_init(self):
        model = crnn.CRNN(32, 1, 37, 256)
        path = self.config.TRAIN.VAL.crnn_pretrained
        if path:
            self.logging.info("loading pretrained crnn model from %s" % path)
            model.load_state_dict(torch.load(path, map_location="cpu"))
        else:
            fill_module_(model)
        model = model.to(self.device).eval()
        for p in model.parameters():
            p.requires_grad = False
        return model, None

    def parse_crnn_data(self, imgs_input):
        return K.bicubic_gray(imgs_input, 100)

    def save_checkpoint(self, netG, epoch, iters, best_acc_dict, best_model_info, is_best, converge_list, exp_name):
        if self.rank != 0:                                  # replicas are bit-identical: rank 0 writes
            return
        os.makedirs(self.ckpt_path, exist_ok=True)
        save_dict = {
            "state_dict_G": {k: v.detach().cpu().contiguous() for k, v in netG.state_dict().items()},
            "info": {"arch": self.args.arch, "iters": iters, "epochs": epoch, "batch_size": self.batch_size,
                     "voc_type": self.voc_type, "up_scale_factor": self.scale_factor},
            "best_history_res": best_acc_dict, "best_model_info": best_model_info,
            "param_num": sum(p.nelement() for p in netG.parameters()), "converge": converge_list}
        torch.save(save_dict, os.path.join(self.ckpt_path, "model_best.pth" if is_best else "checkpoint.pth"))
