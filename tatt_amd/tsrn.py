"""MI355X-native TSRN / TATT generators behind the reference's nn.Module surface.

Drop-in for the reference's ``model/tsrn.py`` classes ``TSRN`` (:88-150) and ``TSRN_TL_TRANS`` (:576-692):
same constructor kwargs, same ``forward`` signatures and return structures, same ``state_dict`` keys/shapes
(304 for TATT with STN), same default initialisation under a given ``torch.manual_seed`` (sub-modules are
constructed in the reference's order; torch.nn layers are used ONLY as parameter/buffer holders and
initialisers -- their ``forward`` is never called).  All arithmetic runs in the hand-written HIP kernels of
libtatt_hip.so (tatt_amd.functional); inputs must live on an AMD GPU, there is no CPU fallback.

Internally feature maps are NHWC; tensors handed back to the caller (SR image, ``block`` dict, ``ret_mid``)
have the reference's logical NCHW shapes (channels-last strides).
"""
from __future__ import annotations

import copy
import math

import numpy as np
import torch
from torch import nn

from . import functional as Fh
from .ops import ACT_NONE, ACT_RELU, ACT_MISH, ACT_TANH
from . import ops


def _nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------------
# parameter holders (names = reference attribute names => identical state_dict keys)
# ---------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the computation runs in tatt_amd's HIP kernels")


class GruBlock(_Holder):
    """reference GruBlock, model/tsrn.py:1067-1084: conv1 (1x1) + bidirectional GRU(out, out/2)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        assert out_channels % 2 == 0
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=1, padding=0)
        self.gru = nn.GRU(out_channels, out_channels // 2, bidirectional=True, batch_first=True)


class RecurrentResidualBlock(_Holder):
    """reference RecurrentResidualBlock (:850-871) / RecurrentResidualBlockTL (:874-910) parameters."""

    def __init__(self, channels, text_channels=0):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(channels)
        self.gru1 = GruBlock(channels + text_channels, channels)
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(channels)
        self.gru2 = GruBlock(channels, channels)


class UpsampleBLock(_Holder):
    """reference UpsampleBLock (:1040-1053): conv 3x3 C -> C*s^2, PixelShuffle(s), mish."""

    def __init__(self, in_channels, up_scale):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels * up_scale ** 2, kernel_size=3, padding=1)


class _EncoderLayer(_Holder):
    def __init__(self, d_model, nhead, dim_ff, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_ff)
        self.linear2 = nn.Linear(dim_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout


class _DecoderLayer(_Holder):
    def __init__(self, d_model, nhead, dim_ff, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)       # unused upstream (:817-819)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_ff)
        self.linear2 = nn.Linear(dim_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)                                              # unused upstream
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.p = dropout


class _Stack(_Holder):
    def __init__(self, layer, n, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
        if norm is not None:
            self.norm = norm


class InfoTransformer(_Holder):
    """reference InfoTransformer (model/transformer_v2.py:154-196) parameters, same construction order."""

    def __init__(self, d_model, nhead, num_encoder_layers, num_decoder_layers, dim_feedforward, dropout,
                 feat_height, feat_width):
        super().__init__()
        self.encoder = _Stack(_EncoderLayer(d_model, nhead, dim_feedforward, dropout), num_encoder_layers)
        dec_layer = _DecoderLayer(d_model, nhead, dim_feedforward, dropout)
        self.decoder = _Stack(dec_layer, num_decoder_layers, nn.LayerNorm(d_model))
        self.gru_encoding = nn.GRU(d_model * feat_height, d_model * feat_height // 2, bidirectional=True,
                                   batch_first=True)
        for p in self.parameters():          # reference _reset_parameters (:193-196)
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.feat_size = (feat_height, feat_width)


def _sincos_table(max_len, d_model):
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


class PositionalEncoding(_Holder):
    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.p = dropout
        self.register_buffer("pe", _sincos_table(max_len, d_model))


class TPInterpreter(_Holder):
    """reference TPInterpreter (model/tsrn.py:155-224) parameters."""

    def __init__(self, t_emb, out_text_channels, output_size=(16, 64), feature_in=64, t_encoder_num=1,
                 t_decoder_num=2):
        super().__init__()
        d_model = out_text_channels
        self.fc_in = nn.Linear(t_emb, d_model)
        self.fc_feature_in = nn.Linear(feature_in, d_model)          # unused upstream, kept for the state_dict
        self.activation = nn.PReLU()
        self.transformer = InfoTransformer(d_model, 4, t_encoder_num, t_decoder_num, d_model, 0.1,
                                           output_size[0], output_size[1])
        self.pe = PositionalEncoding(d_model, 0.1, 5000)
        self.output_size = output_size
        self.seq_len = output_size[0] * output_size[1]
        self.init_factor = nn.Embedding(self.seq_len, d_model)
        # test hook: False reproduces "every nn.Dropout in eval mode, BatchNorm in train mode" (parity runs)
        self.dropout_on = True


def _conv_bn_relu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(cout),
                         nn.ReLU(inplace=True))


class STNHead(_Holder):
    """reference STNHead (model/stn_head.py:25-90) parameters + initialisation."""

    def __init__(self, in_planes, num_ctrlpoints, activation="none", input_size=(16, 64)):
        super().__init__()
        self.num_ctrlpoints = num_ctrlpoints
        self.stn_convnet = nn.Sequential(
            _conv_bn_relu(in_planes, 32), nn.MaxPool2d(2, 2),
            _conv_bn_relu(32, 64), nn.MaxPool2d(2, 2),
            _conv_bn_relu(64, 128), nn.MaxPool2d(2, 2),
            _conv_bn_relu(128, 256), nn.MaxPool2d(2, 2),
            _conv_bn_relu(256, 256), nn.MaxPool2d((1, 2), (1, 2)),
            _conv_bn_relu(256, 256))
        self.stn_fc1 = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True))
        self.stn_fc2 = nn.Linear(512, num_ctrlpoints * 2)
        for seq in (self.stn_convnet, self.stn_fc1):
            for m in seq.modules():
                if isinstance(m, nn.Conv2d):
                    n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                    m.weight.data.normal_(0, math.sqrt(2.0 / n))
                    m.bias.data.zero_()
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.data.fill_(1)
                    m.bias.data.zero_()
                elif isinstance(m, nn.Linear):
                    m.weight.data.normal_(0, 0.001)
                    m.bias.data.zero_()
        margin = 0.01
        k = num_ctrlpoints // 2
        xs = np.linspace(margin, 1.0 - margin, k)
        pts = np.concatenate([np.stack([xs, np.full(k, margin)], 1), np.stack([xs, np.full(k, 1 - margin)], 1)], 0)
        self.stn_fc2.weight.data.zero_()
        self.stn_fc2.bias.data = torch.Tensor(pts.astype(np.float32)).view(-1)


def _tps_phi(a, b):
    """0.5 * d^2 * log(d^2) radial basis between point sets (reference tps_spatial_transformer.py:22-34)."""
    diff = a.view(-1, 1, 2) - b.view(1, -1, 2)
    d2 = diff[:, :, 0] * diff[:, :, 0] + diff[:, :, 1] * diff[:, :, 1]
    r = 0.5 * d2 * torch.log(d2)
    r.masked_fill_(r != r, 0)
    return r


class TPSSpatialTransformer(_Holder):
    """reference TPSSpatialTransformer (model/tps_spatial_transformer.py:54-95) buffers."""

    def __init__(self, output_image_size, num_control_points, margins):
        super().__init__()
        self.output_image_size = output_image_size
        self.num_control_points = num_control_points
        H, W = output_image_size
        N = num_control_points
        k = N // 2
        xs = np.linspace(margins[0], 1.0 - margins[0], k)
        pts = np.concatenate([np.stack([xs, np.full(k, margins[1])], 1),
                              np.stack([xs, np.full(k, 1.0 - margins[1])], 1)], 0)
        tcp = torch.Tensor(pts)
        fk = torch.zeros(N + 3, N + 3)
        fk[:N, :N].copy_(_tps_phi(tcp, tcp))
        fk[:N, -3].fill_(1)
        fk[-3, :N].fill_(1)
        fk[:N, -2:].copy_(tcp)
        fk[-2:, :N].copy_(tcp.transpose(0, 1))
        inverse_kernel = torch.inverse(fk).contiguous()       # (torch.inverse returns column-major strides)
        ys, xs_ = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        coord = torch.stack([xs_.reshape(-1).float(), ys.reshape(-1).float()], 1)
        Y = coord[:, 1:2] / (H - 1)
        X = coord[:, 0:1] / (W - 1)
        coord = torch.cat([X, Y], 1)
        rep = torch.cat([_tps_phi(coord, tcp), torch.ones(H * W, 1), coord], 1)
        self.register_buffer("inverse_kernel", inverse_kernel)
        self.register_buffer("padding_matrix", torch.zeros(3, 2))
        self.register_buffer("target_coordinate_repr", rep)
        self.register_buffer("target_control_points", tcp)


# ---------------------------------------------------------------------------------------------------
# functional forward pieces (HIP)
# ---------------------------------------------------------------------------------------------------
def _require_gpu(x):
    if not x.is_cuda:
        raise RuntimeError("tatt_amd: inputs must be on an AMD GPU (x.device=%s); the product path has no CPU "
                           "fallback (the CPU restatement lives in oracle/ and is test infrastructure)." % x.device)


# the STN head as one launch per layer and direction + one per direction for its fully connected end (csrc/stnhead.hip); False: operator
# by operator (test hook; also taken for geometries the fused launches do not cover)
STN_FUSED = True


def _stn_forward(x_nchw, stn: STNHead, count=True):
    """STNHead.forward (model/stn_head.py:92-106): control points (B, N, 2)."""
    h = x_nchw.permute(0, 2, 3, 1)                    # NHWC-indexed view of the NCHW image
    if STN_FUSED and Fh.stn_head_fusable(h, stn, h.shape[0]):
        if count:
            for bn in [stn.stn_convnet[i][1] for i in (0, 2, 4, 6, 8, 10)] + [stn.stn_fc1[1]]:
                bn.num_batches_tracked += 1
        return Fh.stn_head(h, stn).reshape(h.shape[0], stn.num_ctrlpoints, 2)
    pools = {0: (2, 2), 2: (2, 2), 4: (2, 2), 6: (2, 2), 8: (1, 2)}
    for i in (0, 2, 4, 6, 8, 10):
        conv, bn = stn.stn_convnet[i][0], stn.stn_convnet[i][1]
        h = Fh.conv2d(h, conv.weight, conv.bias)
        h = Fh.batch_norm_act(h, bn, ACT_RELU, count)
        if i in pools:
            h = Fh.max_pool(h, *pools[i])
    B = h.shape[0]
    # x.view(B, -1) of the NCHW map: feature index = c*(H*W) + h*W + w
    h = Fh.Permute4dFn.apply(h, (0, 3, 1, 2)).reshape(B, -1)
    fc1, bn1 = stn.stn_fc1[0], stn.stn_fc1[1]
    h = Fh.linear(h, fc1.weight, fc1.bias)
    h = Fh.batch_norm_act(h, bn1, ACT_RELU, count)
    h = Fh.ScaleFn.apply(h, 0.1)
    h = Fh.linear(h, stn.stn_fc2.weight, stn.stn_fc2.bias)
    return h.reshape(B, stn.num_ctrlpoints, 2)


def _tps_forward(x_nchw, ctrl, tps: TPSSpatialTransformer):
    """TPSSpatialTransformer.forward (model/tps_spatial_transformer.py:97-112) -> NHWC rectified image, src coords."""
    assert ctrl.dim() == 3 and ctrl.size(1) == tps.num_control_points and ctrl.size(2) == 2
    src = Fh.TpsGridFn.apply(ctrl.contiguous(), tps.inverse_kernel, tps.padding_matrix, tps.target_coordinate_repr)
    return Fh.GridSampleFn.apply(x_nchw, src), src


def _gru_block(x, blk: GruBlock, vertical, x_cat=None):
    """GruBlock.forward (model/tsrn.py:1075-1084) on NHWC; `vertical` scans image columns (the reference feeds
    gru1 the H/W-transposed map, :907), x_cat = second half of the channel concat (tp_map)."""
    return Fh.gru_block(x, blk, vertical, xb=x_cat)


def _srb(x, tp_map, blk: RecurrentResidualBlock):
    """RecurrentResidualBlock[TL].forward (model/tsrn.py:862-871, 892-910).  (num_batches_tracked: bumped by the generator.)"""
    x, x_res = Fh.fork2(x)                       # two consumers (conv1 and the residual sum): their gradients meet in one library launch
    if blk.bn1.training and blk.bn2.training and ops.conv3_bn_fusable(x, blk.conv1.weight, blk.bn1) and ops.conv3_bn_fusable(x, blk.conv2.weight, blk.bn2):
        # conv(+stats) | finish | conv(+bn1, mish on the way in, +stats) | finish | apply bn2: 5 launches instead of 8, and the
        # normalised + activated map between the two convolutions never exists in HBM
        if Fh.SRB_BWD_FUSED and ops.CONV3_SB:
            r = Fh.srb_trunk(_cc(x), blk)                        # the same forward; the backward folds both BatchNorm backwards too
        else:
            y1, st1 = Fh.conv_bn(x, blk.conv1, blk.bn1)
            y2, st2 = Fh.conv_bn(y1, blk.conv2, blk.bn2, prev=(st1, blk.bn1, ACT_MISH))
            r = Fh.bn_apply_stats(y2, st2, blk.bn2, ACT_NONE)
    else:
        r = Fh.conv2d(x, blk.conv1.weight, blk.conv1.bias)
        r = Fh.batch_norm_act(r, blk.bn1, ACT_MISH, False)
        r = Fh.conv2d(r, blk.conv2.weight, blk.conv2.bias)
        r = Fh.batch_norm_act(r, blk.bn2, ACT_NONE, False)
    r = _gru_block(r, blk.gru1, True, x_cat=tp_map)
    return _gru_block(Fh.add(x_res, r), blk.gru2, False)


def _ffn(x, layer, training, site):
    h = Fh.linear(x, layer.linear1.weight, layer.linear1.bias, act=ACT_RELU)
    h = Fh.dropout(h, layer.p, training, site)
    return Fh.linear(h, layer.linear2.weight, layer.linear2.bias)


def _query_pos(ig: TPInterpreter, B, H, W):
    """Query positional embedding (model/transformer_v2.py:201-221) -> (B, H*W, C).  Depends on parameters only."""
    C = ig.init_factor.weight.shape[1]
    return Fh.query_embedding(ig.init_factor.weight, ig.transformer.gru_encoding, B, H, W).reshape(B, H * W, C)


def _enc_params(enc):
    sa = enc.self_attn
    return (sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, enc.linear1.weight, enc.linear1.bias,
            enc.linear2.weight, enc.linear2.bias, enc.norm1.weight, enc.norm1.bias, enc.norm2.weight, enc.norm2.bias)


def _dec_params(dec):
    ca = dec.multihead_attn
    return (ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias, dec.linear1.weight, dec.linear1.bias,
            dec.linear2.weight, dec.linear2.bias, dec.norm2.weight, dec.norm2.bias, dec.norm3.weight, dec.norm3.bias)


# Data parallel: file the query GRU (18.9 of 30.4 MB) with the TP interpreter's bucket so that its all-reduce travels while the STN
# head back-propagates?  Measured on one GPU against an RCCL group of one rank (bench.py --dp-selftest, same box, round 3): 7.28 ms
# per step with it, 6.81 without (6.59 without a process group): the 47-launch backward chain of the query GRU then sits beside
# block1's short backward and lengthens that pass by 0.47 ms -- more than the ~0.3 ms the 19 MB all-reduce costs when it trails the
# last pass.  Off by default; tools/ab_bench.py tatt_amd.tsrn.DP_QGRU_WITH_TP=1 re-measures.
DP_QGRU_WITH_TP = False
# where the query GRU's backward (one persistent recurrence launch + four GEMMs since round 4) is filed: "first" = it runs beside the STN
# head's backward (the last pass), "tp" = one pass earlier, beside block1's backward
QGRU_BUCKET = "first"
# where the 9x9 output convolution's weight gradient is filed (see grad_buckets): "srb0" = beside the TP interpreter's backward (rounds
# 2-4), "trunk" = with its own stage (the side lane of the NEXT pass), "now" = its own stage's bucket AND issued at once on the second
# stream, beside the backward from the loss that produced its operands (a pass with no other side work)
OUTCONV_BUCKET = "srb0"
# ... and its bias gradient (a column sum over the 196,608 HR pixels of a 4-channel map): beside the TP layers' backward it waits for
# their work-groups to leave the CUs (156 us in the step, 15 alone) on a side lane that is the longer one of that pass; filed with its
# own stage it runs in a pass whose side lane has room
OUTCONV_BIAS_BUCKET = "trunk"
PRECOMPOSE_ON_FORK = True    # test / A-B hook: False -> the GruBlock compose / pack launches run on the main lane, in front of the STN head
TP_FUSED = True          # test hook: False walks the operator-by-operator path for every geometry (tests compare the two)


def _tp_fusable(ig: TPInterpreter, L):
    """The one-kernel-per-layer path (csrc/tplayer.hip) covers the geometry the reference instantiates (model/tsrn.py:175-182):
    d_model 64, 4 heads, dim_feedforward 64, one encoder layer, one or two decoder layers, at most 32 prior steps."""
    tr = ig.transformer
    enc, decs = tr.encoder.layers, tr.decoder.layers
    if not TP_FUSED:
        return False
    if len(enc) != 1 or not 1 <= len(decs) <= 2 or L > 32:
        return False
    for m, att in [(enc[0], enc[0].self_attn)] + [(d, d.multihead_attn) for d in decs]:
        if att.embed_dim != 64 or att.num_heads != 4 or tuple(m.linear1.weight.shape) != (64, 64):
            return False
    return True


def _tp_text_side(tp, ig: TPInterpreter, training):
    """The text side of the TP interpreter (model/tsrn.py:194-216, transformer_v2.py:268-276): -> (src, pos, memory); memory is None
    when the encoder layer has to run operator by operator (the caller does that)."""
    B, L, C = tp.shape[0], tp.shape[3], ig.fc_in.weight.shape[0]
    x = Fh.Permute4dFn.apply(tp, (0, 3, 2, 1)).reshape(B, L, tp.shape[1])               # (B,26,37); differentiable: the prior may
                                                                                       # come from a trainable recogniser (tatt_amd.crnn)
    x = Fh.prelu(Fh.linear(x, ig.fc_in.weight, ig.fc_in.bias), ig.activation.weight)   # (B,26,64)
    pe = ig.pe.pe[0, :L]                                                               # (26,64)
    tr = ig.transformer
    drop = training and ig.dropout_on
    if drop:
        # pe(zeros) passes through Dropout(0.1) per sample (model/tsrn.py:214; transformer_v2.py:39-42)
        pos = Fh.dropout(pe.unsqueeze(0).expand(B, L, C).contiguous(), ig.pe.p, True, 1)
    else:
        pos = pe
    src = Fh.ScaleFn.apply(x, 2.0)               # the encoder layer is fed with output + src = 2*src (transformer_v2.py:274)
    if not _tp_fusable(ig, L):
        return src, pos, None
    enc = tr.encoder.layers[0]
    pd = lambda v: float(v) if drop else 0.0
    cfg = Fh.TPStackCfg((2,), pd(enc.self_attn.dropout), pd(enc.p), pd(enc.p), False, False, True, enc.norm1.eps)
    memory, _ = Fh.TPStackFn.apply(_cc(src), _cc(pos), src, pos, cfg, *_enc_params(enc))
    return src, pos, memory


def _tp_interpreter(feat, tp, ig: TPInterpreter, training, qpos=None, text=None):
    """TPInterpreter.forward (model/tsrn.py:194-224) + InfoTransformer.forward (model/transformer_v2.py:198-244).
    feat (B,H,W,C) NHWC block1 output; tp (B,37,1,26).  Returns tp_map (B,H,W,C), pr_weights (B,H*W,26)."""
    B, H, W, C = feat.shape
    L = tp.shape[3]
    tr = ig.transformer
    drop = training and ig.dropout_on
    pd = lambda v: float(v) if drop else 0.0
    if text is None:
        text = _tp_text_side(tp, ig, training)
    src, pos, memory = text
    if qpos is None:
        qpos = _query_pos(ig, B, H, W)
    tgt = feat.reshape(B, H * W, C)
    if memory is None:
        return _tp_layers_unfused(src, pos, tgt, qpos, tr, drop, (B, H, W, C))
    Fh.stamp("fwd: tp waits for qpos", feat)
    Fh.FWD_FORK.join(feat.device)              # the query embedding may have been a parallel branch until here
    Fh.stamp("fwd: tp has qpos", feat)
    decs = list(tr.decoder.layers)
    d0 = decs[0]
    cfg = Fh.TPStackCfg([10 + 10 * i for i in range(len(decs))], pd(d0.multihead_attn.dropout), pd(d0.p), pd(d0.p), True, True,
                        False, tr.decoder.norm.eps)
    dparams = [t for d in decs for t in _dec_params(d)] + [tr.decoder.norm.weight, tr.decoder.norm.bias]
    tp_tok, wts = Fh.TPStackFn.apply(_cc(tgt), _cc(qpos), memory, pos, cfg, *dparams)
    return tp_tok.reshape(B, H, W, C), wts


def _cc(t):
    return t if t.is_contiguous() else t.contiguous()


def _tp_layers_unfused(src, pos, tgt, qpos, tr, drop, shape):
    """The encoder / decoder layers operator by operator (any width, head count, depth): ~14 launches per layer."""
    B, H, W, C = shape
    add_pos = Fh.add if pos.dim() == 3 else Fh.AddRowBcastFn.apply
    enc = tr.encoder.layers[0]
    qk = add_pos(src, pos)
    a, _ = Fh.multihead_attention(qk, qk, src, enc.self_attn, drop, 2)
    src = Fh.layer_norm(src, a, enc.norm1, enc.p, drop, 3)
    f = _ffn(src, enc, drop, 4)
    memory = Fh.layer_norm(src, f, enc.norm2, enc.p, drop, 5)
    # decoder: cross-attention only (self-attention commented out upstream, :817-819)
    kmem = add_pos(memory, pos)
    Fh.FWD_FORK.join(tgt.device)              # the query embedding may have been a parallel branch until here
    outs, wts = [], None
    for li, dec in enumerate(tr.decoder.layers):
        s0 = 10 + 10 * li
        a, wts = Fh.multihead_attention(Fh.add(tgt, qpos), kmem, memory, dec.multihead_attn, drop, s0)
        tgt = Fh.layer_norm(tgt, a, dec.norm2, dec.p, drop, s0 + 1)
        f = _ffn(tgt, dec, drop, s0 + 2)
        tgt = Fh.layer_norm(tgt, f, dec.norm3, dec.p, drop, s0 + 3)
        outs.append(Fh.layer_norm(tgt, None, tr.decoder.norm))
    tp_tok = Fh.MeanOf2Fn.apply(outs[0], outs[1]) if len(outs) == 2 else sum(outs) / len(outs)
    return tp_tok.reshape(B, H, W, C), wts


# ---------------------------------------------------------------------------------------------------
# the two generators
# ---------------------------------------------------------------------------------------------------
class _TrainPathMixin:
    """Training-loop plumbing shared by the generators (tatt_amd.train.Trainer drives it; a plain `loss.backward()` loop never
    needs it):

    * `grad_buckets()`: the parameters in the order their gradients complete during the backward pass -- trunk + up-sampler
      (block2..), then the TP interpreter, then block1 + STN head -- one list per bucket of the data-parallel all-reduce;
    * `set_grad_cuts(cuts)`: with a `tatt_amd.dp.GradCuts` installed the forward detaches the tensors that connect those three
      parts, so that the backward can be run stage by stage and bucket k's all-reduce overlaps stage k+1;
    * `_bump_bn_counters()`: ONE launch advances `num_batches_tracked` of every BatchNorm on the path (their buffers are
      re-homed as views of one int64 vector; `state_dict` keys and values are unchanged)."""

    _grad_cuts = None

    def set_grad_cuts(self, cuts):
        object.__setattr__(self, "_grad_cuts", cuts)

    def grad_buckets(self, dp=False):
        """[(stage name, [parameters])] in the order the parameters' gradients are complete; stage names match the
        `cuts.cut(name, ...)` calls of the forward: "trunk" (block7 + up-sampler: what the loss back-propagates into directly),
        "srb4" ... "srb0" (the residual blocks, last to first), "tp" (TP interpreter), "first" (block1 + TPS sampler), "stn" (STN
        head).  A parameter may be filed under a LATER bucket than the stage that produces its gradient: its weight-gradient
        kernels then run with that later stage's side lane (tatt_amd.functional.SIDE.due_of).  Two are: the query GRU (47 dependent
        launches, "first": beside the STN head's backward) and, with a TP interpreter, the 9x9 output convolution (0.4 ms of
        weight gradient, "srb0": beside the TP interpreter's long, launch-bound backward instead of the first residual block's).
        `dp` (data parallel) with DP_QGRU_WITH_TP: the query GRU is filed under "tp" instead (measured slower, see the flag)."""
        k = self.srb_nums
        groups = {"trunk": [], "tp": [], "first": [], "stn": []}
        groups.update({"srb%d" % i: [] for i in range(k)})
        for name, p in self.named_parameters():
            top = name.split(".", 1)[0]
            if top == "infoGen":
                is_q = name.startswith("infoGen.transformer.gru_encoding.") or name.startswith("infoGen.init_factor.")
                groups[QGRU_BUCKET if (is_q and not (dp and DP_QGRU_WITH_TP)) else "tp"].append(p)
            elif top.startswith("block") and 2 <= int(top[5:]) <= k + 1:
                groups["srb%d" % (int(top[5:]) - 2)].append(p)
            elif top.startswith("block") and int(top[5:]) > k + 1:
                out_conv = int(top[5:]) == k + 3 and name.split(".")[1] == str(len(getattr(self, top)) - 1)
                late = out_conv and hasattr(self, "infoGen") and k > 0
                if late and name.endswith(".bias"):
                    groups[OUTCONV_BIAS_BUCKET].append(p)
                else:
                    groups[("trunk" if OUTCONV_BUCKET == "now" else OUTCONV_BUCKET) if late else "trunk"].append(p)
            elif top == "stn_head":
                groups["stn"].append(p)
            else:                                  # block1, (TBSRN's unused conv / bn)
                groups["first"].append(p)
        order = ["trunk"] + ["srb%d" % i for i in range(k - 1, -1, -1)] + ["tp", "first", "stn"]
        return [(n, groups[n]) for n in order if groups[n]]

    def immediate_grad_params(self):
        """Parameters whose gradient kernels run AT ONCE on the side lane of the very pass that produces them (OUTCONV_BUCKET = "now": the
        9x9 output convolution's weight -- its operands exist at the first kernel of the backward, and that pass has no side work).
        (Round 5 tried the same for the STN head's deepest weight gradients in the LAST pass: a dependency from the middle of a main
        lane to the side lane costs the two lanes their overlap, +0.23 ms -- profiles/r05_tail_lane_ab.txt.)"""
        if OUTCONV_BUCKET != "now" or not hasattr(self, "infoGen") or self.srb_nums == 0:
            return []
        b8 = getattr(self, "block%d" % (self.srb_nums + 3))
        return [b8[len(b8) - 1].weight]

    def _bn_on_path(self):
        skip = () if getattr(self, "stn", False) else ("stn_head",)
        return [m for n, m in self.named_modules()
                if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)) and m.track_running_stats and n.split(".", 1)[0] not in skip
                and n != "bn"]                        # TBSRN's unused top-level `bn` never runs

    def _bump_bn_counters(self):
        bns = self.__dict__.get("_bn_list")
        if bns is None:
            bns = self._bn_on_path()
            object.__setattr__(self, "_bn_list", bns)
        if not bns:
            return
        grp = self.__dict__.get("_nbt_group")
        ok = grp is not None and grp.device == bns[0].num_batches_tracked.device and all(
            bn.num_batches_tracked.data_ptr() == grp.data_ptr() + 8 * i for i, bn in enumerate(bns))
        if not ok:
            # (re-)group: happens on the first training forward and after .to(device); never inside a captured step, the Trainer
            # runs eager steps first
            grp = torch.stack([bn.num_batches_tracked.reshape(()) for bn in bns])
            for i, bn in enumerate(bns):
                bn._buffers["num_batches_tracked"] = grp[i]
            object.__setattr__(self, "_nbt_group", grp)
        if all(bn.training for bn in bns):
            if grp.is_cuda:
                ops.inc_i64(grp)
            else:
                grp += 1
        else:                                    # some BatchNorms frozen by the caller: only the live ones count
            for bn in bns:
                if bn.training:
                    bn.num_batches_tracked += 1


class _GeneratorBase(_TrainPathMixin, nn.Module):
    def _build_trunk(self, scale_factor, width, height, STN, srb_nums, mask, hidden_units, text_channels):
        in_planes = 4 if mask else 3
        assert math.log(scale_factor, 2) % 1 == 0
        upsample_block_num = int(math.log(scale_factor, 2))
        C = 2 * hidden_units
        self.block1 = nn.Sequential(nn.Conv2d(in_planes, C, kernel_size=9, padding=4), nn.PReLU())
        self.srb_nums = srb_nums
        for i in range(srb_nums):
            setattr(self, "block%d" % (i + 2), RecurrentResidualBlock(C, text_channels))
        return in_planes, C, upsample_block_num

    def _build_tail(self, in_planes, C, upsample_block_num, scale_factor, width, height, STN, stn_kw):
        srb_nums = self.srb_nums
        setattr(self, "block%d" % (srb_nums + 2), nn.Sequential(nn.Conv2d(C, C, kernel_size=3, padding=1),
                                                                  nn.BatchNorm2d(C)))
        blk = [UpsampleBLock(C, 2) for _ in range(upsample_block_num)]
        blk.append(nn.Conv2d(C, in_planes, kernel_size=9, padding=4))
        setattr(self, "block%d" % (srb_nums + 3), nn.Sequential(*blk))
        self.tps_inputsize = [height // scale_factor, width // scale_factor]
        self.stn = STN
        if self.stn:
            self.tps = TPSSpatialTransformer(tuple(self.tps_inputsize), 20, (0.05, 0.05))
            self.stn_head = STNHead(in_planes, 20, "none", **stn_kw)

    def _check_hip_submodules(self):
        return self

    def _trunk_forward(self, x, text_emb, use_tp):
        _require_gpu(x)
        if x.dtype != torch.float32:
            raise RuntimeError("tatt_amd computes in fp32; got %s" % x.dtype)
        training = self.training
        k = self.srb_nums
        cuts = self._grad_cuts if training else None
        qpos = None
        pre_forked = False
        if use_tp:
            if PRECOMPOSE_ON_FORK and k > 0 and isinstance(getattr(self, "block2").gru1, GruBlock):
                # the composed / packed GruBlock projections are parameters-only work as well, first needed by the first residual
                # block: they lead the forked branch instead of standing in front of the STN head on the main lane (25 us)
                Fh.FWD_FORK.run(x, lambda: Fh.gru_precompose([g for i in range(k) for g in (getattr(self, "block%d" % (i + 2)).gru1,
                                                                                         getattr(self, "block%d" % (i + 2)).gru2)]))
                pre_forked = True
            # first of all: the query embedding depends on parameters only (no dropout in it) and is the longest dependent chain of
            # the forward's head -- its forked branch starts before anything else is issued
            qpos = _query_pos(self.infoGen, x.shape[0], x.shape[2], x.shape[3])
        if training:
            Fh.begin_training_forward(x.device)                  # fresh dropout masks for this call (and its backward)
            self._bump_bn_counters()
        if use_tp and text_emb is None:
            text_emb = torch.zeros(1, 37, 1, 26, device=x.device)         # reference :653-654
        text_side = None
        if not pre_forked and k > 0 and isinstance(getattr(self, "block2").gru1, GruBlock):
            # the composed 1x1-conv x GRU-input projections of every residual block: parameters only, one launch for all of them
            Fh.gru_precompose([g for i in range(k) for g in (getattr(self, "block%d" % (i + 2)).gru1,
                                                             getattr(self, "block%d" % (i + 2)).gru2)])
        if self.stn and training:
            ctrl = _stn_forward(x, self.stn_head, False)
            if cuts:
                ctrl = cuts.cut("stn", ctrl)                     # the STN head's backward is a stage of its own
            Fh.stamp("fwd: stn head done", x)
            xin, _ = _tps_forward(x, ctrl, self.tps)             # NHWC
        else:
            xin = x.permute(0, 2, 3, 1)                          # NHWC-indexed view, read through strides
        c1 = self.block1[0]
        b1 = Fh.prelu(Fh.conv2d(xin, c1.weight, c1.bias), self.block1[1].weight)
        Fh.stamp("fwd: block1 done", x)
        feats = {"1": b1}
        tp_map = pr_weights = None
        b1_trunk = b1
        if use_tp:
            Fh.FWD_FORK_B.join(x.device)             # the text prior may have been a parallel branch until here (TextPriorSR's student)
            tp_map, pr_weights = _tp_interpreter(cuts.cut("first", b1) if cuts else b1, text_emb.float(), self.infoGen, training,
                                                 qpos, text_side)
        tp_ret = tp_map
        if cuts:                                 # backward stages: "trunk" (from the loss), "srb4" ... "srb0", "tp", "first"
            b1_trunk = cuts.cut("first", b1)
        h = b1
        for i in range(k):
            if cuts:                             # the block's inputs: the stream from the block below, its own copy of the prior map
                h_in = cuts.cut("first", b1) if i == 0 else cuts.cut("srb%d" % (i - 1), h)
                tp_in = cuts.cut("tp", tp_map) if tp_map is not None else None
            else:
                h_in, tp_in = h, tp_map
            if i == 0:
                Fh.stamp("fwd: tp done", x)
            h = _srb(h_in, tp_in, getattr(self, "block%d" % (i + 2)))
            Fh.stamp("fwd: srb%d done" % i, x)
            feats[str(i + 2)] = h
        if cuts and k > 0:
            h = cuts.cut("srb%d" % (k - 1), h)
        b7 = getattr(self, "block%d" % (k + 2))
        if b7[1].training and ops.conv3_bn_fusable(h, b7[0].weight, b7[1]):
            y7, st7 = Fh.conv_bn(_cc(h), b7[0], b7[1])
            h = Fh.bn_apply_stats(y7, st7, b7[1], ACT_NONE)
        else:
            h = Fh.conv2d(h, b7[0].weight, b7[0].bias)
            h = Fh.batch_norm_act(h, b7[1], ACT_NONE, False)
        feats[str(k + 2)] = h
        b8 = getattr(self, "block%d" % (k + 3))
        u = Fh.add(b1_trunk, h)
        for m in list(b8)[:-1]:
            u = Fh.conv2d(u, m.conv.weight, m.conv.bias)
            u = Fh.PixelShuffleActFn.apply(u, ACT_MISH)
        last = b8[len(b8) - 1]
        u = Fh.conv2d(u, last.weight, last.bias)
        feats[str(k + 3)] = u
        sr = Fh.ActFn.apply(u, ACT_TANH)                         # reference :675
        Fh.stamp("fwd: sr done", x)
        Fh.gru_precompose_done()                                 # composed projections no block consumed do not outlive the forward
        self.block = {kk: _nchw(v) for kk, v in feats.items()}
        return _nchw(sr), tp_ret, pr_weights, b1


class TSRN(_GeneratorBase):
    """Drop-in for reference ``TSRN`` (model/tsrn.py:88-150)."""

    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32):
        super().__init__()
        in_planes, C, ub = self._build_trunk(scale_factor, width, height, STN, srb_nums, mask, hidden_units, 0)
        self._build_tail(in_planes, C, ub, scale_factor, width, height, STN, {})

    def forward(self, x):
        sr, _, _, _ = self._trunk_forward(x, None, False)
        return sr


class TSRN_TL_TRANS(_GeneratorBase):
    """Drop-in for reference ``TSRN_TL_TRANS`` = TATT (model/tsrn.py:576-692)."""

    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32,
                 word_vec_d=300, text_emb=37, out_text_channels=64, feature_rotate=False, rotate_train=3.):
        super().__init__()
        in_planes, C, ub = self._build_trunk(scale_factor, width, height, STN, srb_nums, mask, hidden_units,
                                             out_text_channels)
        self.infoGen = TPInterpreter(text_emb, out_text_channels,
                                     output_size=(height // scale_factor, width // scale_factor))
        self.feature_rotate = feature_rotate
        self.rotate_train = rotate_train
        self._build_tail(in_planes, C, ub, scale_factor, width, height, STN,
                         {"input_size": [height // scale_factor, width // scale_factor]})
        self.block_range = [k for k in range(2, self.srb_nums + 2)]

    def forward(self, x, text_emb=None, text_emb_gt=None, feature_arcs=None, rand_offs=None):
        sr, tp_map, pr_weights, b1 = self._trunk_forward(x, text_emb, True)
        if self.training:
            tp_nchw = _nchw(tp_map)
            ret_mid = {"pr_weights": pr_weights, "pr_weights_gt": None, "spatial_t_emb": tp_nchw,
                       "spatial_t_emb_gt": None, "in_feat": _nchw(b1), "trans_feat": tp_nchw}
            return sr, ret_mid
        return sr, pr_weights
