"""MI355X-native CRNN text-prior generator behind the reference's nn.Module surface (reference model/crnn/crnn.py:5-92).

SURVEY.md 8f rank 1: the recogniser that runs immediately before the SR generator in every training / inference step and
produces the (B,37,1,26) text prior the TP interpreter consumes (interfaces/super_resolution.py:794-799):

    lr image --bicubic 32x100, luminance--> CRNN: 7 convs (+BN, ReLU, 4 max-pools) -> 26 x B x 512 -> 2 x [BiLSTM(256) + Linear]
             -> logits (26, B, 37) --softmax, permute--> text prior (B, 37, 1, 26)

Same constructor, `forward` signature and state_dict keys / shapes / default initialisation as the reference `CRNN`
(49 keys, 8,331,301 parameters for CRNN(32, 1, 37, 256)); torch.nn layers are parameter holders only, every operator runs as a
HIP kernel of libtatt_hip.so through `tatt_amd.functional` (implicit-GEMM MFMA convolutions, fused BatchNorm+ReLU, general
max-pool, per-step fused LSTM kernels), forward and backward -- the student prior generator is trained through the SR loss.
There is no CPU fallback.
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fh
from . import ops
from .ops import ACT_NONE, ACT_RELU
from .tsrn import _Holder, _require_gpu


class BidirectionalLSTM(_Holder):
    """reference BidirectionalLSTM (model/crnn/crnn.py:5-26): parameters of nn.LSTM(nIn, nHidden, bidirectional) + nn.Linear."""

    def __init__(self, nIn, nHidden, nOut):
        super().__init__()
        self.rnn = nn.LSTM(nIn, nHidden, bidirectional=True)
        self.embedding = nn.Linear(nHidden * 2, nOut)


def _bidirectional_lstm(x, blk: BidirectionalLSTM):
    rec = Fh.bilstm(x, blk.rnn)                                           # (T, B, 2H)
    return Fh.linear(rec, blk.embedding.weight, blk.embedding.bias)       # (T, B, nOut)


class CRNN(nn.Module):
    """Drop-in for reference ``CRNN`` (model/crnn/crnn.py:29-92)."""

    _POOLS = {0: (2, 2, 2, 2, 0, 0), 1: (2, 2, 2, 2, 0, 0), 3: (2, 2, 2, 1, 0, 1), 5: (2, 2, 2, 1, 0, 1)}

    def __init__(self, imgH, nc, nclass, nh, n_rnn=2, leakyRelu=False):
        super().__init__()
        assert imgH % 16 == 0, "imgH has to be a multiple of 16"
        if leakyRelu:
            raise NotImplementedError("the TATT recipes build CRNN(32, 1, 37, 256) with ReLU (interfaces/base.py:713)")
        ks = [3, 3, 3, 3, 3, 3, 2]
        ps = [1, 1, 1, 1, 1, 1, 0]
        nm = [64, 128, 256, 256, 512, 512, 512]
        cnn = nn.Sequential()                         # holder: same registration order / names as the reference
        for i in range(7):
            n_in = nc if i == 0 else nm[i - 1]
            cnn.add_module("conv%d" % i, nn.Conv2d(n_in, nm[i], ks[i], 1, ps[i]))
            if i in (2, 4, 6):
                cnn.add_module("batchnorm%d" % i, nn.BatchNorm2d(nm[i]))
            cnn.add_module("relu%d" % i, nn.ReLU(True))
            if i in (0, 1):
                cnn.add_module("pooling%d" % i, nn.MaxPool2d(2, 2))
            elif i == 3:
                cnn.add_module("pooling2", nn.MaxPool2d((2, 2), (2, 1), (0, 1)))
            elif i == 5:
                cnn.add_module("pooling3", nn.MaxPool2d((2, 2), (2, 1), (0, 1)))
        self.cnn = cnn
        self.rnn = nn.Sequential(BidirectionalLSTM(512, nh, nh), BidirectionalLSTM(nh, nh, nclass))

    def forward(self, input):
        """input (B, nc, 32, W) fp32 on the GPU -> logits (W/4 + 1, B, nclass)."""
        _require_gpu(input)
        h = input.permute(0, 2, 3, 1)                                     # NHWC view (a plain reshape when nc == 1)
        for i in range(7):
            conv = getattr(self.cnn, "conv%d" % i)
            bn = getattr(self.cnn, "batchnorm%d" % i, None)
            if conv.kernel_size == (3, 3):
                h = Fh.conv2d(h, conv.weight, conv.bias, ACT_NONE if bn is not None else ACT_RELU, any_width=True)   # (split-bf16 3x3 kernels on the 50 / 25 / 26-pixel maps)
            else:
                h = Fh.Conv2x2ValidFn.apply(Fh._c(h), conv.weight, conv.bias)
                if bn is None:
                    h = Fh.ActFn.apply(h, ACT_RELU)
            if bn is not None:
                h = Fh.batch_norm_act(h, bn, ACT_RELU)
            if i in self._POOLS:
                h = Fh.max_pool(h, *self._POOLS[i])
        B, Hh, Wd, C = h.shape
        assert Hh == 1, "the height of conv must be 1"
        seq = Fh.Permute4dFn.apply(h, (2, 1, 0, 3)).reshape(Wd, B, C)     # (W', B, 512) time-major
        seq = _bidirectional_lstm(seq, self.rnn[0])
        return _bidirectional_lstm(seq, self.rnn[1])


class LumaResizeFn(torch.autograd.Function):
    """parse_crnn_data (reference interfaces/base.py:797-815): bicubic resize of img[:, :3] to (32, in_width) + luminance.
    The image is data (the reference detaches it, super_resolution.py:786): no gradient."""

    @staticmethod
    def forward(ctx, img, oh, ow):
        ops._check_dev(img)
        B, C, H, W = img.shape
        assert C >= 3
        out = ops.new(img, B, 1, oh, ow)
        ops.call("tatt_bicubic_luma", ops.P(img), *img.stride(), ops.P(out), B, H, W, oh, ow, ops.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return None, None, None


def parse_crnn_data(img, in_width=100):
    """(B, >=3, H, W) image on the GPU -> (B, 1, 32, in_width) recogniser input."""
    return LumaResizeFn.apply(img.detach(), 32, in_width)


def text_prior(logits):
    """(T, B, 37) logits -> (B, 37, 1, T) softmax prior for `TSRN_TL_TRANS.forward(x, text_emb)` (reference
    interfaces/super_resolution.py:796-799).  Softmax over 37 classes of 26 x B rows: the fused row-softmax kernel."""
    T, B, C = logits.shape
    p = Fh.SoftmaxRowsFn.apply(Fh._c(logits).reshape(T * B, C)).reshape(T, B, C)
    return p.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
