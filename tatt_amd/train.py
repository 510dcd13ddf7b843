"""Training-step harness for the TATT hot path: loss, global-norm clip, Adam, data parallelism.

Mirrors the reference's hot loop (interfaces/super_resolution.py:873-894,1072-1085):
    sr = model(lr, text_prior); loss = ImageLoss(sr, hr).mean()*100; zero_grad; backward;
    clip_grad_norm_(0.25); Adam(1e-3, betas=(0.5, 0.999)).step()
and replaces torch.nn.DataParallel (interfaces/base.py:386-396) by one process per GPU with an RCCL
all-reduce of ONE flat gradient buffer over xGMI (`torch.distributed`, backend "nccl" = RCCL on ROCm).

* Parameters, gradients and Adam moments live in flat fp32 buffers (7.6 M elements = 30.4 MB each); the module's
  nn.Parameters are views into the flat parameter buffer, their `.grad`s views into the flat gradient buffer.
* Clip + Adam are two HIP kernels (tatt_l2norm, tatt_adam_step) whose step-varying scalars live in device
  memory, so a whole step (forward, loss, backward, optimiser) can be captured once as a hipGraph and replayed.
* ImageLoss (loss/image_loss.py) is one fused forward and one fused backward HIP kernel (tatt_image_loss_*) instead of
  ~80 element-wise torch kernels on the (B,4,2H,2W) images.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import functional as Fh
from . import ops
from .dp import FlatParams, broadcast_model, allreduce_grads


def image_loss(sr, hr, weights=(1.0, 1e-4)):
    """ImageLoss(gradient=True, loss_weight=[1, 1e-4]).forward (reference loss/image_loss.py:19-34): per-sample loss (B,)."""
    return Fh.ImageLossFn.apply(sr, hr, float(weights[0]), float(weights[1]), None)


def image_loss_mean(sr, hr, weights=(1.0, 1e-4), scale=100.0):
    """`ImageLoss(sr, hr).mean() * scale` of the training loop (interfaces/super_resolution.py:889-894) as one scalar."""
    return Fh.ImageLossFn.apply(sr, hr, float(weights[0]), float(weights[1]), float(scale))


def semantic_loss(pred, gt):
    """SemanticLoss()(pred, gt) of the reference (loss/semantic_loss.py:21-38): prior distillation, student vs teacher."""
    return Fh.SemanticLossFn.apply(pred, gt.detach())


def calculate_psnr(img1, img2):
    """calculate_psnr of the reference (utils/ssim_psnr.py:9-15): device scalar, images in [0, 1], first 3 channels."""
    ops._check_dev(img1)
    ops._check_dev(img2)
    B, C, H, W = img1.shape
    out = ops.new(img1, 1)
    ops.call("tatt_psnr", ops.P(img1), *img1.stride(), ops.P(img2), *img2.stride(), ops.P(out), B, C, H, W, ops.stream())
    return out.reshape(())


class TextPriorSR(torch.nn.Module):
    """The generator together with its trainable text-prior generator, as the reference's loop composes them
    (interfaces/super_resolution.py:786-815): lr image -> parse_crnn_data -> CRNN student -> softmax prior -> SR(x, prior).
    Gradients of the SR loss reach the recogniser through the prior.  With a frozen `teacher` recogniser the step also carries
    the prior-distillation term of the reference, `sem_loss(student prior on LR, teacher prior on HR) * 100`
    (super_resolution.py:767,879): `Trainer` adds `extra_loss(hr)` to the image loss."""

    def __init__(self, sr, tpg, teacher=None, in_width=100):
        super().__init__()
        self.sr, self.tpg, self.in_width = sr, tpg, in_width
        object.__setattr__(self, "_teacher", teacher)        # frozen: deliberately NOT a registered sub-module / parameter owner
        self._student_probs = None

    def _apply(self, fn, *args, **kwargs):                 # .to(device) moves the (unregistered) teacher too
        super()._apply(fn, *args, **kwargs)
        if self._teacher is not None:
            self._teacher._apply(fn, *args, **kwargs)
        return self

    @property
    def block(self):
        return self.sr.block

    @block.setter
    def block(self, v):
        self.sr.block = v

    def _probs(self, net, img):
        from .crnn import parse_crnn_data
        logits = net(parse_crnn_data(img[:, :3], self.in_width))                 # (T, B, 37)
        T, B, C = logits.shape
        return Fh.SoftmaxRowsFn.apply(logits.reshape(T * B, C)).reshape(T, B, C)

    def forward(self, x):
        probs = self._probs(self.tpg, x)
        self._student_probs = probs
        prior = probs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)           # (B, 37, 1, T)
        return self.sr(x, prior)

    def extra_loss(self, hr):
        """Distillation term; None without a teacher.  Call after forward()."""
        if self._teacher is None:
            return None
        with torch.no_grad():
            gt = self._probs(self._teacher, hr)
        loss = semantic_loss(self._student_probs, gt) * 100.0
        self._student_probs = None
        return loss


class Trainer:
    """One training step per `step()` call; optional whole-step hipGraph; optional data parallelism."""

    def __init__(self, model, lr=1e-3, betas=(0.5, 0.999), eps=1e-8, clip=0.25, use_graph=False, warmup_eager=2,
                 process_group=None, broadcast_init=True):
        self.model = model
        self.lr, self.betas, self.eps, self.clip = lr, betas, eps, clip
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.flat = FlatParams(model)
        self.params = self.flat.params
        self.n = self.flat.n
        dev = self.flat.p.device
        self.dev = dev
        self.flat_p, self.flat_g = self.flat.p, self.flat.g
        self.flat_m = torch.zeros(self.n, device=dev)
        self.flat_v = torch.zeros(self.n, device=dev)
        if self.world > 1 and broadcast_init:
            broadcast_model(self.flat, model, process_group)
        self.gnorm = torch.zeros(1, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.norm_ws = torch.empty(1024, dtype=torch.float64, device=dev)
        self.use_graph = use_graph
        self.warmup_eager = warmup_eager
        self._graphs = None
        self._static = None
        self._nsteps = 0
        self.last_loss = None

    # -- pieces ----------------------------------------------------------------------------------
    def _fwd_bwd(self, x, tp, hr):
        # gradients are produced as fresh tensors by the HIP backward kernels and gathered into the flat buffer with one
        # multi-tensor copy (instead of ~290 per-parameter `grad += new` kernels through pre-assigned .grad views)
        for p in self.params:
            p.grad = None
        out = self.model(x, tp) if tp is not None else self.model(x)
        sr = out[0] if isinstance(out, tuple) else out
        loss = image_loss_mean(sr, hr, scale=100.0)
        extra = self.model.extra_loss(hr) if hasattr(self.model, "extra_loss") else None
        if extra is not None:
            loss = loss + extra
        loss.backward()
        self.model.block = None                      # do not keep the autograd graph of this step alive
        self.flat_g.zero_()
        have = [p for p in self.params if p.grad is not None]
        torch._foreach_copy_([self.flat.grad_view(p) for p in have], [p.grad for p in have])
        return loss.detach()

    def _optim(self):
        self.step_count += 1
        ops.call("tatt_l2norm", ops.P(self.flat_g), self.n, ops.P(self.gnorm), ops.P(self.norm_ws), ops.stream())
        ops.call("tatt_adam_step", ops.P(self.flat_p), ops.P(self.flat_g), ops.P(self.flat_m), ops.P(self.flat_v),
                 self.n, self.lr, self.betas[0], self.betas[1], self.eps, ops.P(self.gnorm), self.clip,
                 1.0 / self.world, ops.P(self.step_count), ops.stream())
        Fh.next_dropout_step(self.dev)

    def _allreduce(self):
        if self.world > 1:
            allreduce_grads(self.flat, self.pg)       # sum; the 1/world factor is folded into tatt_adam_step

    @property
    def last_grad_norm(self):
        """||g||_2 of the (rank-averaged) gradient before clipping."""
        return self.gnorm / self.world

    # -- public ----------------------------------------------------------------------------------
    def step(self, x, tp, hr):
        """x (B,4,H,W), tp (B,37,1,26) or None (TSRN), hr (B,4,2H,2W), all on this rank's GPU.  Returns the loss
        tensor (device scalar, no host sync)."""
        self._nsteps += 1
        if not self.use_graph or self._nsteps <= self.warmup_eager:
            loss = self._fwd_bwd(x, tp, hr)
            self._allreduce()
            self._optim()
            self.last_loss = loss
            return loss
        if self._graphs is None:
            self._capture(x, tp, hr)
        sx, stp, shr = self._static
        sx.copy_(x)
        shr.copy_(hr)
        if stp is not None:
            stp.copy_(tp)
        g1, g2 = self._graphs
        g1.replay()
        if g2 is not None:
            self._allreduce()
            g2.replay()
        return self.last_loss

    def _capture(self, x, tp, hr):
        self._static = (x.clone(), None if tp is None else tp.clone(), hr.clone())
        sx, stp, shr = self._static
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        g2 = None
        with torch.cuda.graph(g1):
            self.last_loss = self._fwd_bwd(sx, stp, shr)
            if self.world == 1:
                self._optim()
        if self.world > 1:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                self._optim()
        self._graphs = (g1, g2)
