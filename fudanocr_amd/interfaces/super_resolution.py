"""TextSR: train / eval / test loops of the reference (interfaces/super_resolution.py:37-329) on the HIP
path.  One optimisation step = engine.TrainStep (model.train() -> sr -> crit -> (loss*100).backward() ->
[all-reduce] -> clip 0.25 -> Adam), eval = SR -> PSNR/SSIM -> parse_crnn_data -> CRNN -> greedy decode ->
exact-match accuracy."""
import os
import time
from datetime import datetime

import torch

from ..utils import ssim_psnr
from ..utils.util import str_filt
from ..utils.utils_crnn import get_crnn_pred
from . import base


class TextSR(base.TextBase):
    def train(self):
        cfg = self.config.TRAIN
        train_dataset, train_loader = self.get_train_data()
        val_dataset_list, val_loader_list = self.get_val_data()
        md = self.generator_init()
        model, image_crit, rec = md["model"], md["crit"], md["recognizer"]
        step = self.optimizer_init(model, image_crit)
        best_history_acc, best_model_acc, best_acc, converge_list = {}, {}, 0, []
        t_start, n_img = time.time(), 0
        for epoch in range(cfg.epochs):
            sampler = getattr(train_loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)      # DistributedSampler: a new permutation per epoch (ShardLoader advances itself)
            for j, data in enumerate(train_loader):
                iters = len(train_loader) * epoch + j
                images_hr, images_lr, label_strs = data
                out = step(images_lr.to(self.device), images_hr.to(self.device), label_strs)
                n_img += images_lr.shape[0]
                if iters % cfg.displayInterval == 0:
                    self.logging.info("[{}]\tEpoch: [{}][{}/{}]\ttotal_loss {:.3f} \tmse_loss {:.3f} \tctc_loss {:.3f} \t"
                                      "{:.1f} img/s".format(datetime.now().strftime("%Y-%m-%d %H:%M:%S"), epoch, j + 1,
                                                            len(train_loader), float(out["loss"]) * 100, float(out["mse"]),
                                                            float(out["ctc"]) if out["ctc"] is not None else -1,
                                                            n_img / (time.time() - t_start)))
                if iters % cfg.VAL.valInterval == 0 and iters > 0 or iters == len(train_loader) * cfg.epochs - 1:
                    self.logging.info("======================================================")
                    current_acc = {}
                    for k, val_loader in enumerate(val_loader_list):
                        name = "val%d" % k
                        m = self.eval(model, val_loader, image_crit, iters, rec)
                        converge_list.append({"iterator": iters, "acc": m["accuracy"], "psnr": m["psnr_avg"],
                                              "ssim": m["ssim_avg"]})
                        current_acc[name] = float(m["accuracy"])
                        if current_acc[name] >= best_history_acc.get(name, -1):
                            best_history_acc[name] = current_acc[name]
                    if sum(current_acc.values()) >= best_acc:
                        best_acc, best_model_acc = sum(current_acc.values()), current_acc
                        self.save_checkpoint(model, epoch, iters, best_history_acc, best_model_acc, True, converge_list,
                                             self.exp_name)
                if iters % cfg.saveInterval == 0:
                    self.save_checkpoint(model, epoch, iters, best_history_acc, best_model_acc, False, converge_list,
                                         self.exp_name)
        return {"images_per_sec": n_img / (time.time() - t_start), "best_acc": best_history_acc}

    def get_crnn_pred(self, outputs):
        return get_crnn_pred(outputs)

    @torch.no_grad()
    def eval(self, model, val_loader, image_crit, index, recognizer):
        model.eval()
        n_correct, sum_images, psnr, ssim = 0, 0, [], []
        for data in val_loader:
            images_hr, images_lr, label_strs = data
            images_lr, images_hr = images_lr.to(self.device), images_hr.to(self.device)
            images_sr = model(images_lr)
            p_, s_ = ssim_psnr.psnr_ssim(images_sr, images_hr)      # one fused device pass (cal_psnr / cal_ssim
            psnr.append(float(p_))                                  # give the same values one metric at a time)
            ssim.append(float(s_))
            if recognizer is not None:
                out = recognizer(self.parse_crnn_data(images_sr[:, :3])).permute(1, 0, 2).contiguous()
                pred = self.get_crnn_pred(out)
                n_correct += sum(p == str_filt(t, "lower") for p, t in zip(pred, label_strs))
            sum_images += images_lr.shape[0]
        acc = round(n_correct / max(sum_images, 1), 4)
        res = {"accuracy": acc, "psnr_avg": round(sum(psnr) / len(psnr), 6), "ssim_avg": round(sum(ssim) / len(ssim), 6)}
        self.logging.info("[{}]\tPSNR {:.2f} | SSIM {:.4f}\taccuracy {:.2f}%".format(
            datetime.now().strftime("%Y-%m-%d %H:%M:%S"), res["psnr_avg"], res["ssim_avg"], acc * 100))
        return res

    def test(self):
        md = self.generator_init()
        tdir = getattr(self.args, "test_data_dir", "") or ""
        if tdir and os.path.isdir(tdir):          # reference test(): one loader per sub-directory of --test_data_dir
            subs = [os.path.join(tdir, d) for d in sorted(os.listdir(tdir)) if os.path.isdir(os.path.join(tdir, d))]
            loaders = [self.get_test_data(d)[1] for d in (subs or [tdir])]
        else:
            _, loaders = self.get_val_data()
        t0 = time.time()
        res = [self.eval(md["model"], ld, md["crit"], 0, md["recognizer"]) for ld in loaders]
        n = sum(len(ld) for ld in loaders) * self.batch_size
        self.logging.info("fps %.1f" % (n / (time.time() - t0)))
        return res

    @torch.no_grad()
    def demo(self):
        """Reference interfaces/super_resolution.py:331-420: every image of --demo_dir is resized to 64 x 16 (PIL
        bicubic), optionally given the mean-threshold mask channel, super-resolved, and recognised by the CRNN both
        from the LR input and from the SR output; logs '<lr string> ===> <sr string>' per image and the fps."""
        import numpy as np
        from PIL import Image
        mask_ = self.mask

        def transform_(path):
            img = Image.open(path)
            img = img.resize((64, 16), Image.BICUBIC)
            arr = np.asarray(img.convert("RGB"), dtype=np.uint8)
            t = torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div_(255.0)       # ToTensor
            if mask_:
                m = img.convert("L")
                thres = np.array(m).mean()
                m = m.point(lambda x: 0 if x > thres else 255)
                t = torch.cat((t, torch.from_numpy(np.asarray(m, dtype=np.uint8).copy())[None].float().div_(255.0)), 0)
            return t.unsqueeze(0)

        md = self.generator_init()
        model, rec = md["model"], md["recognizer"]
        if rec is None:
            rec, _ = self.CRNN_init()
        rec.eval()
        for p in model.parameters():
            p.requires_grad = False
        model.eval()
        names = sorted(os.listdir(self.args.demo_dir))
        results, sr_time, t0 = [], 0.0, time.time()
        for im_name in names:
            images_lr = transform_(os.path.join(self.args.demo_dir, im_name)).to(self.device)
            t1 = time.time()
            images_sr = model(images_lr)
            sr_time += time.time() - t1
            strs = []
            for im in (images_lr, images_sr):
                # parse_crnn_data resizes BOTH axes to (32, 100) (base.py:319-325): the LR branch goes 16 -> 32 rows
                # first (eval-only resampling on the device), then the shared HIP bicubic-luma kernel along W
                x3 = im[:, :3]
                if x3.shape[2] != 32:
                    x3 = torch.nn.functional.interpolate(x3, (32, x3.shape[3]), mode="bicubic", align_corners=False)
                out = rec(self.parse_crnn_data(x3.contiguous()))                         # [26, 1, 37]
                _, preds = out.max(2)
                preds = preds.transpose(1, 0).contiguous().view(-1)
                strs.append(self.converter_crnn.decode(preds.cpu().to(torch.int32),
                                                       torch.IntTensor([out.size(0)]), raw=False))
            self.logging.info("{} ===> {}".format(strs[0], strs[1]))
            results.append((im_name, strs[0], strs[1]))
        fps = len(names) / max(time.time() - t0, 1e-9)
        self.logging.info("fps={}".format(fps))
        return {"results": results, "fps": fps, "sr_time": sr_time}
