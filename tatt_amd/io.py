"""Checkpoint files and the evaluation metrics loop in the reference's formats.

* `save_checkpoint` writes what `TextBase.save_checkpoint` writes (reference interfaces/base.py:621-672): one file per generator
  with the dict {'state_dict_G', 'info', 'best_history_res', 'best_model_info', 'param_num', 'converge'} as
  `model_best_<prefix>_<i>.pth` / `checkpoint.pth`, the recognisers' state_dicts beside them.
* `load_generator` mirrors the resume branch of `TextBase.generator_init` (reference interfaces/base.py:398-443): a file or a
  directory, with or without the 'state_dict_G' wrapper, `module.`-prefixed keys of DataParallel checkpoints accepted in both
  directions.
* `evaluate` is the metric part of the reference's eval loop (interfaces/super_resolution.py:1409-1420,1454-1455): PSNR and SSIM
  of the SR images against HR on the first three channels, averaged over batches -- computed by the HIP kernels.
torch.save / torch.load are file-format plumbing; all arithmetic stays in the HIP path.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Sequence

import torch


def _unwrap(m):
    return m.module if hasattr(m, "module") and isinstance(getattr(m, "module"), torch.nn.Module) else m


def save_checkpoint(netG_list: Sequence[torch.nn.Module], epoch: int, iters: int, best_acc_dict, best_model_info, is_best: bool,
                    converge_list, ckpt_path: str, *, arch: str = "tatt", batch_size: int = 48, voc_type: str = "all",
                    scale_factor: int = 2, recognizer=None, prefix: str = "acc"):
    """reference TextBase.save_checkpoint (interfaces/base.py:621-672); returns the list of files written."""
    os.makedirs(ckpt_path, exist_ok=True)
    written = []
    for i, net in enumerate(netG_list):
        netG = _unwrap(net)
        save_dict = {
            "state_dict_G": {k: v.detach().cpu() for k, v in netG.state_dict().items()},
            "info": {"arch": arch, "iters": iters, "epochs": epoch, "batch_size": batch_size, "voc_type": voc_type,
                     "up_scale_factor": scale_factor},
            "best_history_res": best_acc_dict,
            "best_model_info": best_model_info,
            "param_num": sum(p.nelement() for p in netG.parameters()),
            "converge": converge_list,
        }
        name = ("model_best_%s_%d.pth" % (prefix, i)) if is_best else "checkpoint.pth"
        torch.save(save_dict, os.path.join(ckpt_path, name))
        written.append(os.path.join(ckpt_path, name))
    if recognizer is not None:
        recs = recognizer if isinstance(recognizer, (list, tuple)) else [recognizer]
        for i, r in enumerate(recs):
            if isinstance(recognizer, (list, tuple)):
                name = ("recognizer_best_%s_%d.pth" % (prefix, i)) if is_best else "recognizer_%d.pth" % i
            else:
                name = "recognizer_best.pth" if is_best else "recognizer.pth"
            torch.save({k: v.detach().cpu() for k, v in _unwrap(r).state_dict().items()}, os.path.join(ckpt_path, name))
            written.append(os.path.join(ckpt_path, name))
    return written


def _strip_module(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_generator(model: torch.nn.Module, resume: str, iter_: int = 0, strict: Optional[bool] = None):
    """reference TextBase.generator_init resume branch (interfaces/base.py:398-443).  `resume` is a checkpoint file or a directory
    holding model_best_acc_<iter_>.pth; directories load with strict=False like the reference, files strictly.  Returns the
    checkpoint's 'info' dict (or None for a bare state_dict)."""
    is_dir = os.path.isdir(resume)
    path = os.path.join(resume, "model_best_acc_%d.pth" % iter_) if is_dir else resume
    blob = torch.load(path, map_location="cpu")
    sd = blob["state_dict_G"] if isinstance(blob, dict) and "state_dict_G" in blob else blob
    target = _unwrap(model)
    target.load_state_dict(_strip_module(sd), strict=(not is_dir) if strict is None else strict)
    return blob.get("info") if isinstance(blob, dict) and "state_dict_G" in blob else None


@torch.no_grad()
def evaluate(model: torch.nn.Module, batches: Iterable, prior_fn=None):
    """Metric part of the reference's eval loop: for every (images_lr, images_hr[, text_prior]) batch run the generator in eval
    mode and accumulate calculate_psnr / SSIM of SR vs HR on the first three channels (interfaces/super_resolution.py:1454-1455),
    plus the same for the LR input when its size matches HR.  Returns {'psnr', 'ssim', 'n_batches'} (python floats)."""
    from .losses import SSIM, calculate_psnr
    was_training = model.training
    model.eval()
    ssim = SSIM()
    psnr_sum = torch.zeros((), device=next(model.parameters()).device)
    ssim_sum = torch.zeros_like(psnr_sum)
    n = 0
    for batch in batches:
        lr, hr = batch[0], batch[1]
        tp = batch[2] if len(batch) > 2 else (prior_fn(lr) if prior_fn is not None else None)
        out = model(lr, tp) if tp is not None else model(lr)
        sr = out[0] if isinstance(out, tuple) else out
        psnr_sum += calculate_psnr(sr[:, :3], hr[:, :3])
        ssim_sum += ssim(sr[:, :3], hr[:, :3])
        n += 1
    model.train(was_training)
    return {"psnr": float(psnr_sum) / max(n, 1), "ssim": float(ssim_sum) / max(n, 1), "n_batches": n}
