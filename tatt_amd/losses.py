"""The losses and the rotation augmentation around the SR generator, as the reference's training loop uses them
 -- same class names, constructor arguments and call signatures, HIP kernels underneath:

    ImageLoss(gradient=True, loss_weight=[1, 1e-4])      reference loss/image_loss.py:9-34
    SemanticLoss()                                       reference loss/semantic_loss.py:7-38
    SSIM(window_size=11, size_average=True)              reference utils/ssim_psnr.py:202-228   (first 3 channels)
    TRI_SSIM(window_size=11, size_average=True)          reference utils/ssim_psnr.py:231-256   (all channels)
    calculate_psnr(img1, img2)                           reference utils/ssim_psnr.py:9-15
    torch_distortion(images, arcs, rand_offs)            reference model/__init__.py:4-29 (= TextSR.torch_rotate_img)

No CPU fallback: tensors must live on an AMD GPU.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import functional as Fh
from .train import calculate_psnr, image_loss, semantic_loss  # noqa: F401  (re-exported)


class ImageLoss(nn.Module):
    def __init__(self, gradient=True, loss_weight=(20, 1e-4)):
        super().__init__()
        self.gradient = gradient
        self.loss_weight = list(loss_weight)

    def forward(self, out_images, target_images, grad_mask=None):
        # grad_mask: accepted and ignored, exactly as the reference does (loss/image_loss.py:19-23 never reads it)
        w1 = float(self.loss_weight[1]) if self.gradient else 0.0
        return image_loss(out_images, target_images, (float(self.loss_weight[0]), w1))


class SemanticLoss(nn.Module):
    def forward(self, pred_vec, gt_vec):
        return semantic_loss(pred_vec, gt_vec)


class _SsimBase(nn.Module):
    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        if window_size != 11:
            # documented limit: every call site of the reference constructs SSIM() / TRI_SSIM() with the default
            raise NotImplementedError("the HIP SSIM kernels implement the 11x11 window every reference call site uses")
        self.window_size, self.size_average = window_size, size_average

    def _reduce(self, per_sample):
        # size_average: mean over every element == mean of the per-sample means (equal counts); else the reference's
        # .mean(1).mean(1).mean(1) per sample
        return per_sample.mean() if self.size_average else per_sample


class SSIM(_SsimBase):
    def forward(self, img1, img2):
        return self._reduce(Fh.SsimFn.apply(img1[:, :3], img2[:, :3], None))


class TRI_SSIM(_SsimBase):
    def forward(self, img1, img2, img3):
        return self._reduce(Fh.SsimFn.apply(img1, img2, img3))


def rotation_theta(arcs: torch.Tensor, rand_offs: torch.Tensor, H: int, W: int, off_range: float = 0.2) -> torch.Tensor:
    """The (N,2,3) affine matrices of torch_distortion (reference model/__init__.py:9-26): rotation by `arcs` with the aspect
    ratio H/W jittered by `rand_offs`.  Plain torch arithmetic on whatever device the inputs live on (N numbers)."""
    arcs, rand_offs = arcs.float(), rand_offs.float()
    rm = H / float(W) + rand_offs * off_range * 2 - off_range
    cos, sin, zero = torch.cos(arcs), torch.sin(arcs), torch.zeros_like(arcs)
    return torch.stack([cos, sin * rm, zero, -sin / rm, cos, zero], 1).reshape(-1, 2, 3)


def torch_distortion(torch_image_batches, arc_batches, rand_offs, off_range=0.2):
    N, C, H, W = torch_image_batches.shape
    theta = rotation_theta(arc_batches, rand_offs, H, W, off_range)
    return Fh.AffineSampleFn.apply(torch_image_batches, theta)


