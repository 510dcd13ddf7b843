"""CPU oracle for the CRNN text-prior generator (the step immediately before the SR hot path).

TEST INFRASTRUCTURE ONLY -- same rules as oracle/tatt_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product (tatt_amd.crnn) never does.

Restates, in plain fp32 PyTorch with explicit arithmetic (own bicubic filter, own max-pool windows, own LSTM cell loop):
  * `parse_crnn_data`   reference interfaces/base.py:797-815   (bicubic resize of the LR image to 32 x 100, luminance)
  * `CRNN.forward`      reference model/crnn/crnn.py:29-92      (7 convs + BN/ReLU/max-pools -> 26 x B x 512 -> 2 BiLSTM(256))
  * the prior fed to the SR model: softmax over the 37 classes, (B,37,1,26)  (interfaces/super_resolution.py:794-799)
Pinned against the reference itself by tools/gen_golden.py (tests/golden/crnn_b2.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .tatt_oracle import batch_norm, conv2d

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def _cubic_weights(t: Tensor, a: float = -0.75):
    """Keys' cubic convolution coefficients for the 4 taps at offsets -1, 0, 1, 2 (ATen upsample_bicubic2d, A = -0.75)."""
    def near(x):      # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1
    def far(x):       # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    return [far(t + 1), near(t), near(1 - t), far(2 - t)]


def bicubic_resize(x: Tensor, oh: int, ow: int) -> Tensor:
    """F.interpolate(x, (oh, ow), mode='bicubic', align_corners=False): source coordinate (o + 0.5) * in/out - 0.5, border taps
    clamped to the image (reference interfaces/base.py:807)."""
    def axis(n_in, n_out):
        src = (torch.arange(n_out, dtype=torch.float32) + 0.5) * (n_in / n_out) - 0.5
        i0 = torch.floor(src)
        t = src - i0
        idx = [torch.clamp(i0.long() + k, 0, n_in - 1) for k in (-1, 0, 1, 2)]
        return idx, _cubic_weights(t)
    ih, iw = x.shape[-2:]
    hi, hw = axis(ih, oh)
    wi, ww = axis(iw, ow)
    rows = sum(x[..., hi[k], :] * hw[k].reshape(-1, 1) for k in range(4))          # (..., oh, iw)
    return sum(rows[..., wi[k]] * ww[k] for k in range(4))                          # (..., oh, ow)


def parse_crnn_data(img: Tensor, in_width: int = 100) -> Tensor:
    """RGB image (B,>=3,H,W) -> (B,1,32,in_width) luminance of the bicubic resize (reference interfaces/base.py:797-815)."""
    r = bicubic_resize(img[:, :3], 32, in_width)
    return 0.299 * r[:, 0:1] + 0.587 * r[:, 1:2] + 0.114 * r[:, 2:3]


def max_pool2d(x: Tensor, k, s, p) -> Tensor:
    """nn.MaxPool2d(k, s, p): windows padded with -inf (reference model/crnn/crnn.py:57-69)."""
    xp = F.pad(x, (p[1], p[1], p[0], p[0]), value=float("-inf"))
    win = xp.unfold(2, k[0], s[0]).unfold(3, k[1], s[1])                             # (B,C,Ho,Wo,kh,kw)
    return win.reshape(*win.shape[:4], -1).max(-1).values


def lstm_direction(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor, reverse: bool) -> Tensor:
    """One direction of nn.LSTM (time-major x (T,B,I) -> (T,B,H)), h0 = c0 = 0; gate order (i, f, g, o):
    c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c')."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t() + b_ih
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = gi[t] + h @ w_hh.t() + b_hh
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, 0)


def bidirectional_lstm(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """BidirectionalLSTM.forward -- reference model/crnn/crnn.py:5-26: BiLSTM then Linear on every (t, b)."""
    p = prefix + ".rnn."
    f = lstm_direction(x, sd[p + "weight_ih_l0"], sd[p + "weight_hh_l0"], sd[p + "bias_ih_l0"], sd[p + "bias_hh_l0"], False)
    r = lstm_direction(x, sd[p + "weight_ih_l0_reverse"], sd[p + "weight_hh_l0_reverse"], sd[p + "bias_ih_l0_reverse"],
                       sd[p + "bias_hh_l0_reverse"], True)
    rec = torch.cat([f, r], -1)
    return rec @ sd[prefix + ".embedding.weight"].t() + sd[prefix + ".embedding.bias"]


def crnn_forward(sd: SD, x: Tensor, *, training: bool = False, new_stats: Optional[dict] = None) -> Tensor:
    """CRNN(32, 1, 37, 256).forward -- reference model/crnn/crnn.py:78-92.  x (B,1,32,W) -> logits (W/4+1, B, 37)."""
    bn = {2, 4, 6}
    pools = {0: ((2, 2), (2, 2), (0, 0)), 1: ((2, 2), (2, 2), (0, 0)), 3: ((2, 2), (2, 1), (0, 1)), 5: ((2, 2), (2, 1), (0, 1))}
    h = x
    for i in range(7):
        w = sd["cnn.conv%d.weight" % i]
        h = conv2d(h, w, sd["cnn.conv%d.bias" % i], 1 if w.shape[-1] == 3 else 0)
        if i in bn:
            h = batch_norm(h, sd, "cnn.batchnorm%d" % i, training, new_stats=new_stats)
        h = torch.relu(h)
        if i in pools:
            h = max_pool2d(h, *pools[i])
    assert h.shape[2] == 1, "the height of conv must be 1"
    seq = h.squeeze(2).permute(2, 0, 1)                                  # (W', B, 512)
    seq = bidirectional_lstm(seq, sd, "rnn.0")
    return bidirectional_lstm(seq, sd, "rnn.1")


def text_prior(logits: Tensor) -> Tensor:
    """(T,B,37) logits -> the (B,37,1,T) prior the SR model consumes (reference interfaces/super_resolution.py:796-799)."""
    return torch.softmax(logits, -1).permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
