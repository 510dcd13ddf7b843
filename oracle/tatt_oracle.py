"""CPU oracle for the TATT / TSRN super-resolution hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch fp32 CPU *restatement*
of the reference algorithm (mjq11302010044/TATT, ``model/tsrn.py`` and friends).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it -- as the checker, never as the product.  The shipped
path (``tatt_amd``) never imports this module and raises when its HIP library is
missing.

Pinning: the reference ships no tests / golden vectors, so
this oracle is pinned against the reference *itself*, imported in the build
container by ``tools/gen_golden.py``; the resulting vectors are committed under
``tests/golden/`` and checked by ``tests/test_oracle_golden.py``.

Every function takes tensors / a reference-format ``state_dict`` (``sd``, same 304
keys as ``TSRN_TL_TRANS.state_dict()`` of the reference) and is written with
explicit arithmetic (own GRU cell loop, own multi-head attention, own layer /
batch norm, own pixel shuffle, own TPS + bilinear sampler); ``F.conv2d`` /
``matmul`` are the only composite primitives.  All functions are differentiable
through ``torch.autograd`` so the same code yields reference gradients.

Reference citations are ``file:line`` into the upstream repository.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# elementary pieces
# ----------------------------------------------------------------------------
def mish(x: Tensor) -> Tensor:
    """``x * tanh(softplus(x))`` -- model/tsrn.py:1056-1064 (softplus beta=1, threshold=20)."""
    sp = torch.where(x > 20.0, x, torch.log1p(torch.exp(torch.clamp(x, max=20.0))))
    return x * torch.tanh(sp)


def prelu(x: Tensor, alpha: Tensor) -> Tensor:
    """nn.PReLU() with ONE shared slope -- model/tsrn.py:598, :173."""
    return torch.where(x >= 0, x, alpha.reshape(()) * x)


def conv2d(x: Tensor, w: Tensor, b: Optional[Tensor], pad: int) -> Tensor:
    """Stride-1 'same' convolution (nn.Conv2d) -- model/tsrn.py:597,877,885,612,623,1043,1071."""
    return F.conv2d(x, w, b, stride=1, padding=pad)


def batch_norm(x: Tensor, sd: SD, prefix: str, training: bool, eps: float = 1e-5,
               momentum: float = 0.1, new_stats: Optional[dict] = None) -> Tensor:
    """nn.BatchNorm2d / BatchNorm1d -- model/tsrn.py:878,886,613; model/stn_head.py:19,51.

    training: normalise with the batch mean and *biased* variance over every axis
    but the channel axis; running stats are updated with the *unbiased* variance
    (momentum 0.1) and ``num_batches_tracked += 1`` -- returned through
    ``new_stats`` (functional: ``sd`` is not mutated).  eval: running stats.
    """
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dims)
        var = ((x - mean.reshape(shape)) ** 2).mean(dims)
        if new_stats is not None:
            with torch.no_grad():
                unb = var * (n / max(n - 1, 1))
                new_stats[prefix + ".running_mean"] = (1 - momentum) * sd[prefix + ".running_mean"] + momentum * mean
                new_stats[prefix + ".running_var"] = (1 - momentum) * sd[prefix + ".running_var"] + momentum * unb
                new_stats[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    xhat = (x - mean.reshape(shape)) / torch.sqrt(var.reshape(shape) + eps)
    return xhat * w.reshape(shape) + b.reshape(shape)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm over the last axis (biased variance) -- model/transformer_v2.py:459-460,793-795."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def pixel_shuffle2(x: Tensor) -> Tensor:
    """nn.PixelShuffle(2): out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w] -- model/tsrn.py:1045."""
    n, c4, h, w = x.shape
    c = c4 // 4
    return x.reshape(n, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, c, 2 * h, 2 * w)


def dropout(x: Tensor, p: float, on: bool) -> Tensor:
    """nn.Dropout.  The oracle only ever runs with ``on=False`` for parity (SURVEY.md §7
    "Dropout in train mode"); ``on=True`` uses torch's RNG and exists for the CPU timing leg."""
    return F.dropout(x, p, training=True) if (on and p > 0) else x


# ----------------------------------------------------------------------------
# GRU (own cell loop)
# ----------------------------------------------------------------------------
def gru_direction(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor,
                  reverse: bool) -> Tensor:
    """One direction of nn.GRU(batch_first=True), h0 = 0.  x: (N, T, I) -> (N, T, H).

    Gate order (r, z, n); n = tanh(W_in x + b_in + r * (W_hn h + b_hn));
    h' = (1 - z) * n + z * h.  (torch.nn.GRU semantics relied on by
    model/tsrn.py:1072 and model/transformer_v2.py:177.)
    """
    N, T, _ = x.shape
    H = w_hh.shape[1]
    gi_all = x @ w_ih.t() + b_ih                       # (N, T, 3H)
    h = x.new_zeros(N, H)
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gi = gi_all[:, t]
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 1)


def bigru(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """Bidirectional single-layer GRU: concat(forward, reverse) on the feature axis."""
    f = gru_direction(x, sd[prefix + ".weight_ih_l0"], sd[prefix + ".weight_hh_l0"],
                      sd[prefix + ".bias_ih_l0"], sd[prefix + ".bias_hh_l0"], False)
    r = gru_direction(x, sd[prefix + ".weight_ih_l0_reverse"], sd[prefix + ".weight_hh_l0_reverse"],
                      sd[prefix + ".bias_ih_l0_reverse"], sd[prefix + ".bias_hh_l0_reverse"], True)
    return torch.cat([f, r], -1)


def gru_block(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """GruBlock -- model/tsrn.py:1067-1084.  x (B,Cin,D2,D3): 1x1 conv, then a BiGRU
    scanning the LAST spatial axis (D3) for each of the B*D2 rows."""
    x = conv2d(x, sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"], 0)
    x = x.permute(0, 2, 3, 1)
    b = x.shape
    y = bigru(x.reshape(b[0] * b[1], b[2], b[3]), sd, prefix + ".gru")
    return y.reshape(b[0], b[1], b[2], b[3]).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------
# multi-head attention (own implementation of nn.MultiheadAttention forward)
# ----------------------------------------------------------------------------
def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, sd: SD, prefix: str, nhead: int,
        p_drop: float = 0.1, drop_on: bool = False) -> Tuple[Tensor, Tensor]:
    """nn.MultiheadAttention (batch-major here: q (B,L,E), k/v (B,S,E)).

    packed in-projection rows [q; k; v]; q scaled by 1/sqrt(E/nhead); softmax over S;
    dropout on the probabilities; out-projection; returned weights are the
    (post-dropout) probabilities averaged over heads -- model/transformer_v2.py:453,786,821-824.
    """
    B, L, E = q_in.shape
    S = k_in.shape[1]
    d = E // nhead
    w, bias = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = q_in @ w[:E].t() + bias[:E]
    k = k_in @ w[E:2 * E].t() + bias[E:2 * E]
    v = v_in @ w[2 * E:].t() + bias[2 * E:]
    q = q * (1.0 / math.sqrt(d))
    q = q.reshape(B, L, nhead, d).permute(0, 2, 1, 3)
    k = k.reshape(B, S, nhead, d).permute(0, 2, 1, 3)
    v = v.reshape(B, S, nhead, d).permute(0, 2, 1, 3)
    att = torch.softmax(q @ k.transpose(-1, -2), -1)           # (B,h,L,S)
    att = dropout(att, p_drop, drop_on)
    ctx = (att @ v).permute(0, 2, 1, 3).reshape(B, L, E)
    out = ctx @ sd[prefix + ".out_proj.weight"].t() + sd[prefix + ".out_proj.bias"]
    return out, att.mean(1)


# ----------------------------------------------------------------------------
# TP interpreter (text-prior cross-attention transformer)
# ----------------------------------------------------------------------------
def positional_encoding(L: int, d_model: int) -> Tensor:
    """Sin/cos table -- model/transformer_v2.py:29-37 (the registered buffer ``pe``)."""
    pe = torch.zeros(L, d_model)
    position = torch.arange(0, L).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def query_embedding(sd: SD, prefix: str, B: int, H: int, W: int) -> Tensor:
    """Query positional embedding -- model/transformer_v2.py:201-221.

    ``init_factor.weight`` (H*W, C) is repeated over the batch, reshaped to
    (W, B, H*C) and fed to ``gru_encoding`` (batch_first=True): the GRU therefore sees
    batch = W image columns and *sequence = the B samples* (SURVEY.md §8a-7 quirk),
    with the identical input at every step.  Returns (B, H*W, C).
    """
    emb = sd[prefix + ".init_factor.weight"]                   # (H*W, C)
    C = emb.shape[1]
    x = emb.reshape(H, W, C).permute(1, 0, 2).reshape(W, 1, H * C).expand(W, B, H * C)
    y = bigru(x, sd, prefix + ".transformer.gru_encoding")     # (W, B, H*C)
    y = y.reshape(W, B, H, C).permute(1, 2, 0, 3)              # (B, H, W, C)
    return y.reshape(B, H * W, C)


def encoder_layer(src: Tensor, pos: Tensor, sd: SD, prefix: str, drop_on: bool) -> Tensor:
    """TransformerEncoderLayer.forward_post -- model/transformer_v2.py:470-484 (batch-major)."""
    a, _ = mha(src + pos, src + pos, src, sd, prefix + ".self_attn", 4, 0.1, drop_on)
    src = layer_norm(src + dropout(a, 0.1, drop_on), sd[prefix + ".norm1.weight"], sd[prefix + ".norm1.bias"])
    f = torch.relu(src @ sd[prefix + ".linear1.weight"].t() + sd[prefix + ".linear1.bias"])
    f = dropout(f, 0.1, drop_on) @ sd[prefix + ".linear2.weight"].t() + sd[prefix + ".linear2.bias"]
    return layer_norm(src + dropout(f, 0.1, drop_on), sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"])


def decoder_layer(tgt: Tensor, memory: Tensor, pos: Tensor, query_pos: Tensor, sd: SD, prefix: str,
                  drop_on: bool) -> Tuple[Tensor, Tensor]:
    """TransformerDecoderLayer_TP.forward_post -- model/transformer_v2.py:806-833.
    The self-attention branch is commented out upstream (:817-819): only cross-attention."""
    a, wts = mha(tgt + query_pos, memory + pos, memory, sd, prefix + ".multihead_attn", 4, 0.1, drop_on)
    tgt = layer_norm(tgt + dropout(a, 0.1, drop_on), sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"])
    f = torch.relu(tgt @ sd[prefix + ".linear1.weight"].t() + sd[prefix + ".linear1.bias"])
    f = dropout(f, 0.1, drop_on) @ sd[prefix + ".linear2.weight"].t() + sd[prefix + ".linear2.bias"]
    tgt = layer_norm(tgt + dropout(f, 0.1, drop_on), sd[prefix + ".norm3.weight"], sd[prefix + ".norm3.bias"])
    return tgt, wts


def tp_interpreter(feat: Tensor, tp: Tensor, sd: SD, prefix: str = "infoGen",
                   drop_on: bool = False) -> Tuple[Tensor, Tensor]:
    """TPInterpreter.forward -- model/tsrn.py:194-224 (+ InfoTransformer.forward,
    model/transformer_v2.py:198-244; TransformerEncoder :248-280; TransformerDecoder :346-392).

    feat (B,C,H,W) = block1 output (decoder ``tgt``); tp (B,37,1,26) text prior.
    Returns tp_map (B,C,H,W) and pr_weights (B,H*W,26).
    """
    B, C, H, W = feat.shape
    x = tp.permute(0, 3, 1, 2).squeeze(-1)                                   # (B,26,37)
    x = prelu(x @ sd[prefix + ".fc_in.weight"].t() + sd[prefix + ".fc_in.bias"],
              sd[prefix + ".activation.weight"])                              # (B,26,64)
    L = x.shape[1]
    pos = dropout(sd[prefix + ".pe.pe"][0, :L].unsqueeze(0).expand(B, L, C), 0.1, drop_on)
    tgt = feat.reshape(B, C, H * W).permute(0, 2, 1)                          # (B,P,C)
    qpos = query_embedding(sd, prefix, B, H, W)
    t = prefix + ".transformer"
    # encoder: ONE layer whose input is output + src_item = 2*src  (transformer_v2.py:274)
    memory = encoder_layer(x + x, pos, sd, t + ".encoder.layers.0", drop_on)
    outs = []
    wts = None
    for l in range(2):
        tgt, wts = decoder_layer(tgt, memory, pos, qpos, sd, t + ".decoder.layers.%d" % l, drop_on)
        outs.append(layer_norm(tgt, sd[t + ".decoder.norm.weight"], sd[t + ".decoder.norm.bias"]))
    tp_tok = (outs[0] + outs[1]) * 0.5                                        # .mean(0), tsrn.py:219
    tp_map = tp_tok.permute(0, 2, 1).reshape(B, C, H, W)
    return tp_map, wts


# ----------------------------------------------------------------------------
# STN head + TPS sampler
# ----------------------------------------------------------------------------
def max_pool(x: Tensor, kh: int, kw: int) -> Tensor:
    n, c, h, w = x.shape
    return x.reshape(n, c, h // kh, kh, w // kw, kw).amax((3, 5))


def stn_head(x: Tensor, sd: SD, prefix: str, training: bool, new_stats: Optional[dict]) -> Tensor:
    """STNHead.forward -- model/stn_head.py:92-106.  Returns control points (B,20,2)."""
    pools = {0: (2, 2), 2: (2, 2), 4: (2, 2), 6: (2, 2), 8: (1, 2)}
    for i in (0, 2, 4, 6, 8, 10):
        p = "%s.stn_convnet.%d" % (prefix, i)
        x = conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 1)
        x = torch.relu(batch_norm(x, sd, p + ".1", training, new_stats=new_stats))
        if i in pools:
            x = max_pool(x, *pools[i])
    x = x.reshape(x.shape[0], -1)
    x = x @ sd[prefix + ".stn_fc1.0.weight"].t() + sd[prefix + ".stn_fc1.0.bias"]
    x = torch.relu(batch_norm(x, sd, prefix + ".stn_fc1.1", training, new_stats=new_stats))
    x = (0.1 * x) @ sd[prefix + ".stn_fc2.weight"].t() + sd[prefix + ".stn_fc2.bias"]
    return x.reshape(-1, 20, 2)


def grid_sample_bilinear(inp: Tensor, grid: Tensor) -> Tensor:
    """F.grid_sample(input, grid): bilinear, zeros padding, align_corners=False --
    model/tps_spatial_transformer.py:11.  inp (B,C,H,W); grid (B,Ho,Wo,2) in [-1,1]."""
    B, C, H, W = inp.shape
    gx, gy = grid[..., 0], grid[..., 1]
    ix = ((gx + 1.0) * W - 1.0) * 0.5
    iy = ((gy + 1.0) * H - 1.0) * 0.5
    x0, y0 = torch.floor(ix), torch.floor(iy)
    out = 0
    flat = inp.reshape(B, C, H * W)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wgt = (1.0 - (ix - xi).abs()) * (1.0 - (iy - yi).abs())
            valid = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(B, 1, -1).expand(B, C, -1)
            val = torch.gather(flat, 2, idx).reshape(B, C, *gx.shape[1:])
            out = out + val * (wgt * valid).unsqueeze(1)
    return out


def tps_transform(x: Tensor, ctrl: Tensor, sd: SD, prefix: str) -> Tuple[Tensor, Tensor]:
    """TPSSpatialTransformer.forward -- model/tps_spatial_transformer.py:97-112."""
    B, _, H, W = x.shape
    Y = torch.cat([ctrl, sd[prefix + ".padding_matrix"].expand(B, 3, 2)], 1)
    mapping = sd[prefix + ".inverse_kernel"] @ Y
    src = sd[prefix + ".target_coordinate_repr"] @ mapping                     # (B,H*W,2)
    grid = 2.0 * src.reshape(B, H, W, 2).clamp(0, 1) - 1.0
    return grid_sample_bilinear(x, grid), src


# ----------------------------------------------------------------------------
# residual blocks and the two generators
# ----------------------------------------------------------------------------
def srb(x: Tensor, tp_map: Optional[Tensor], sd: SD, prefix: str, training: bool,
        new_stats: Optional[dict]) -> Tensor:
    """RecurrentResidualBlockTL.forward (model/tsrn.py:892-910) when ``tp_map`` is given,
    RecurrentResidualBlock.forward (:862-871) otherwise."""
    r = conv2d(x, sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"], 1)
    r = mish(batch_norm(r, sd, prefix + ".bn1", training, new_stats=new_stats))
    r = conv2d(r, sd[prefix + ".conv2.weight"], sd[prefix + ".conv2.bias"], 1)
    r = batch_norm(r, sd, prefix + ".bn2", training, new_stats=new_stats)
    if tp_map is not None:
        r = torch.cat([r, tp_map], 1)
    r = gru_block(r.transpose(-1, -2), sd, prefix + ".gru1").transpose(-1, -2)
    return gru_block(x + r, sd, prefix + ".gru2")


def generator_forward(sd: SD, x: Tensor, text_emb: Optional[Tensor] = None, *, training: bool = False,
                      tatt: bool = True, stn: bool = True, srb_nums: int = 5,
                      drop_on: bool = False, new_stats: Optional[dict] = None) -> Dict[str, Tensor]:
    """TSRN_TL_TRANS.forward (model/tsrn.py:646-692) when ``tatt`` else TSRN.forward (:132-150).

    Returns a dict with 'sr' and the intermediates the reference exposes
    (pr_weights, tp_map, block1..block8, ctrl, src_coord)."""
    out: Dict[str, Tensor] = {}
    if stn and training:
        ctrl = stn_head(x, sd, "stn_head", training, new_stats)
        x, src = tps_transform(x, ctrl, sd, "tps")
        out["ctrl"], out["src_coord"], out["x_rect"] = ctrl, src, x
    b1 = prelu(conv2d(x, sd["block1.0.weight"], sd["block1.0.bias"], 4), sd["block1.1.weight"])
    out["block1"] = b1
    tp_map = None
    if tatt:
        if text_emb is None:
            text_emb = torch.zeros(1, 37, 1, 26)
        tp_map, wts = tp_interpreter(b1, text_emb, sd, "infoGen", drop_on)
        out["tp_map"], out["pr_weights"] = tp_map, wts
    h = b1
    for i in range(srb_nums):
        h = srb(h, tp_map, sd, "block%d" % (i + 2), training, new_stats)
        out["block%d" % (i + 2)] = h
    k = srb_nums + 2
    h = conv2d(h, sd["block%d.0.weight" % k], sd["block%d.0.bias" % k], 1)
    h = batch_norm(h, sd, "block%d.1" % k, training, new_stats=new_stats)
    out["block%d" % k] = h
    u = conv2d(b1 + h, sd["block%d.0.conv.weight" % (k + 1)], sd["block%d.0.conv.bias" % (k + 1)], 1)
    u = mish(pixel_shuffle2(u))
    u = conv2d(u, sd["block%d.1.weight" % (k + 1)], sd["block%d.1.bias" % (k + 1)], 4)
    out["block%d" % (k + 1)] = u
    out["sr"] = torch.tanh(u)
    return out


# ----------------------------------------------------------------------------
# TBSRN variant (reference model/tbsrn.py) -- SURVEY.md §8a-16
# ----------------------------------------------------------------------------
def positionalencoding2d(d_model: int, height: int, width: int) -> Tensor:
    """model/tbsrn.py:39-61: first half of the channels encode the column, second half the row."""
    pe = torch.zeros(d_model, height, width)
    half = d_model // 2
    div_term = torch.exp(torch.arange(0., half, 2) * -(math.log(10000.0) / half))
    pos_w = torch.arange(0., width).unsqueeze(1)
    pos_h = torch.arange(0., height).unsqueeze(1)
    pe[0:half:2] = torch.sin(pos_w * div_term).t().unsqueeze(1).repeat(1, height, 1)
    pe[1:half:2] = torch.cos(pos_w * div_term).t().unsqueeze(1).repeat(1, height, 1)
    pe[half::2] = torch.sin(pos_h * div_term).t().unsqueeze(2).repeat(1, 1, width)
    pe[half + 1::2] = torch.cos(pos_h * div_term).t().unsqueeze(2).repeat(1, 1, width)
    return pe


def tbsrn_layer_norm(x: Tensor, a: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """The variant's own LayerNorm -- model/tbsrn.py:23-36: UNBIASED std, eps added to the std (outside the root)."""
    mean = x.mean(-1, keepdim=True)
    std = torch.sqrt(((x - mean) ** 2).sum(-1, keepdim=True) / (x.shape[-1] - 1))
    return a * (x - mean) / (std + eps) + b


def feature_enhancer(feat: Tensor, sd: SD, prefix: str, drop_on: bool = False) -> Tensor:
    """FeatureEnhancer.forward -- model/tbsrn.py:77-93 (MultiHeadedAttention :96-128, attention :130-151,
    PositionwiseFeedForward :154-164).  feat (B,64,H,W) -> (B,64,H,W)."""
    B, C, H, W = feat.shape
    Pn = H * W
    # reference: positionalencoding2d(64,16,256) viewed as (1,64,4096) -- only defined for H*W == 4096; other sizes use (H, W)
    pe = positionalencoding2d(64, 16, 256).reshape(1, 64, 4096) if Pn == 4096 else positionalencoding2d(64, H, W).reshape(1, 64, Pn)
    x = torch.cat([feat.reshape(B, C, Pn), pe.expand(B, 64, Pn)], 1).permute(0, 2, 1)          # (B,P,128)
    h, d = 4, 32
    lin = lambda i, t: t @ sd["%s.multihead.linears.%d.weight" % (prefix, i)].t() + sd["%s.multihead.linears.%d.bias" % (prefix, i)]
    q, k, v = (lin(i, x).reshape(B, Pn, h, d).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(d), -1)
    p = dropout(p, 0.1, drop_on)
    a = lin(3, (p @ v).transpose(1, 2).reshape(B, Pn, h * d))
    x = tbsrn_layer_norm(x + a, sd[prefix + ".mul_layernorm1.a_2"], sd[prefix + ".mul_layernorm1.b_2"])
    f = torch.relu(x @ sd[prefix + ".pff.w_1.weight"].t() + sd[prefix + ".pff.w_1.bias"])
    f = dropout(f, 0.1, drop_on) @ sd[prefix + ".pff.w_2.weight"].t() + sd[prefix + ".pff.w_2.bias"]
    x = tbsrn_layer_norm(x + f, sd[prefix + ".mul_layernorm3.a_2"], sd[prefix + ".mul_layernorm3.b_2"])
    x = x @ sd[prefix + ".linear.weight"].t() + sd[prefix + ".linear.bias"]
    return x.permute(0, 2, 1).reshape(B, C, H, W)


def tbsrn_forward(sd: SD, x: Tensor, *, training: bool = False, stn: bool = True, srb_nums: int = 5,
                  drop_on: bool = False, new_stats: Optional[dict] = None) -> Dict[str, Tensor]:
    """TBSRN.forward -- model/tbsrn.py:215-227, with RecurrentResidualBlock.forward :366-377."""
    out: Dict[str, Tensor] = {}
    if stn and training:
        ctrl = stn_head(x, sd, "stn_head", training, new_stats)
        x, _ = tps_transform(x, ctrl, sd, "tps")
    b1 = prelu(conv2d(x, sd["block1.0.weight"], sd["block1.0.bias"], 4), sd["block1.1.weight"])
    h = b1
    for i in range(srb_nums):
        pre = "block%d" % (i + 2)
        r = conv2d(h, sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"], 1)
        r = mish(batch_norm(r, sd, pre + ".bn1", training, new_stats=new_stats))
        r = conv2d(r, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], 1)
        r = batch_norm(r, sd, pre + ".bn2", training, new_stats=new_stats)
        h = h + feature_enhancer(r, sd, pre + ".feature_enhancer", drop_on)
        out[pre] = h
    k = srb_nums + 2
    h = conv2d(h, sd["block%d.0.weight" % k], sd["block%d.0.bias" % k], 1)
    h = batch_norm(h, sd, "block%d.1" % k, training, new_stats=new_stats)
    u = conv2d(b1 + h, sd["block%d.0.conv.weight" % (k + 1)], sd["block%d.0.conv.bias" % (k + 1)], 1)
    u = mish(pixel_shuffle2(u))
    u = conv2d(u, sd["block%d.1.weight" % (k + 1)], sd["block%d.1.bias" % (k + 1)], 4)
    out["block1"], out["sr"] = b1, torch.tanh(u)
    return out


# ----------------------------------------------------------------------------
# train-step harness (loss / clip / Adam) -- SURVEY.md §8a-17
# ----------------------------------------------------------------------------
def gradient_map(x: Tensor) -> Tensor:
    """GradientPriorLoss.gradient_map -- loss/image_loss.py:50-58."""
    h, w = x.shape[-2:]
    r = F.pad(x, (0, 1, 0, 0))[:, :, :, 1:]
    l = F.pad(x, (1, 0, 0, 0))[:, :, :, :w]
    t = F.pad(x, (0, 0, 1, 0))[:, :, :h, :]
    b = F.pad(x, (0, 0, 0, 1))[:, :, 1:, :]
    return torch.sqrt(((r - l) * 0.5) ** 2 + ((t - b) * 0.5) ** 2 + 1e-6)


def image_loss(sr: Tensor, hr: Tensor, weights=(1.0, 1e-4)) -> Tensor:
    """ImageLoss.forward (gradient=True) -- loss/image_loss.py:19-34; per-sample loss (B,)."""
    mse = ((sr - hr) ** 2).mean((1, 2, 3))
    gp = (gradient_map(sr[:, :3]) - gradient_map(hr[:, :3])).abs().mean((1, 2, 3))
    return weights[0] * mse + weights[1] * gp


def semantic_loss(pred: Tensor, gt: Tensor) -> Tensor:
    """SemanticLoss.forward -- loss/semantic_loss.py:21-38: L1 + nn.KLDivLoss() (reduction 'mean' over every element)."""
    l1 = (gt - pred).abs().mean()
    t = gt + 1e-20
    kl = (t * (torch.log(t) - torch.log(pred + 1e-20))).mean()
    return l1 + kl


def calculate_psnr(img1: Tensor, img2: Tensor) -> Tensor:
    """utils/ssim_psnr.py:9-15."""
    mse = ((img1[:, :3] * 255 - img2[:, :3] * 255) ** 2).mean()
    return 20 * torch.log10(255.0 / torch.sqrt(mse))


# ----------------------------------------------------------------------------
# SURVEY.md 8f-2 (next row, oracle side only so far): SSIM / TRI_SSIM losses and the rotation augmentation of the shipped recipe
# ----------------------------------------------------------------------------
def gaussian_window(size: int = 11, sigma: float = 1.5) -> Tensor:
    """create_window -- utils/ssim_psnr.py:28-37: normalised 1-D Gaussian, outer product -> (size, size)."""
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    g = g / g.sum()
    return torch.outer(g, g).float()


def _local_mean(x: Tensor, win: Tensor) -> Tensor:
    """Depthwise 'same' correlation with zero padding: F.conv2d(x, window, padding=ws//2, groups=C) (utils/ssim_psnr.py:77)."""
    C = x.shape[1]
    return F.conv2d(x, win.expand(C, 1, *win.shape).contiguous(), padding=win.shape[0] // 2, groups=C)


def ssim(img1: Tensor, img2: Tensor, size_average: bool = True) -> Tensor:
    """SSIM.forward + _ssim -- utils/ssim_psnr.py:212-228,76-97: on the first 3 channels only (the mask channel is dropped)."""
    img1, img2 = img1[:, :3], img2[:, :3]
    win = gaussian_window()
    mu1, mu2 = _local_mean(img1, win), _local_mean(img2, win)
    s1 = _local_mean(img1 * img1, win) - mu1 * mu1
    s2 = _local_mean(img2 * img2, win) - mu2 * mu2
    s12 = _local_mean(img1 * img2, win) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean((1, 2, 3))


def tri_ssim(img1: Tensor, img2: Tensor, img3: Tensor, size_average: bool = True) -> Tensor:
    """_tri_ssim -- utils/ssim_psnr.py:99-129 (used as (1 - tri_ssim(sr_rot, sr, hr).mean()) * 10, super_resolution.py:910-914)."""
    win = gaussian_window()
    mu = [_local_mean(i, win) for i in (img1, img2, img3)]
    sq = [_local_mean(i * i, win) - m * m for i, m in zip((img1, img2, img3), mu)]
    cross = lambda a, b, ma, mb: _local_mean(a * b, win) - ma * mb               # noqa: E731
    s12, s23, s31 = cross(img1, img2, mu[0], mu[1]), cross(img2, img3, mu[1], mu[2]), cross(img3, img1, mu[2], mu[0])
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((mu[0] * mu[1] + mu[1] * mu[2] + mu[2] * mu[0] + C1) * (s12 + s23 + s31 + C2)) / \
        ((mu[0] ** 2 + mu[1] ** 2 + mu[2] ** 2 + C1) * (sq[0] + sq[1] + sq[2] + C2))
    return m.mean() if size_average else m.mean((1, 2, 3))


def affine_grid(theta: Tensor, H: int, W: int) -> Tensor:
    """F.affine_grid(theta (N,2,3), (N,C,H,W)), align_corners=False: base coordinates (2i+1)/size - 1."""
    xs = (2 * torch.arange(W, dtype=torch.float32) + 1) / W - 1
    ys = (2 * torch.arange(H, dtype=torch.float32) + 1) / H - 1
    base = torch.stack([xs.expand(H, W), ys.unsqueeze(1).expand(H, W), torch.ones(H, W)], -1)      # (H,W,3)
    return torch.einsum("hwk,njk->nhwj", base, theta)


def torch_distortion(img: Tensor, arcs: Tensor, rand_offs: Tensor, off_range: float = 0.2) -> Tensor:
    """torch_distortion / TextSR.torch_rotate_img -- model/__init__.py:4-29, interfaces/super_resolution.py:126-157: rotation by
    `arcs` with the aspect ratio jittered by `rand_offs`, bilinear resampling with zero padding."""
    N, C, H, W = img.shape
    rm = H / float(W) + rand_offs * off_range * 2 - off_range
    cos, sin, zero = torch.cos(arcs), torch.sin(arcs), torch.zeros_like(arcs)
    theta = torch.stack([cos, sin * rm, zero, -sin / rm, cos, zero], 1).reshape(N, 2, 3)
    return grid_sample_bilinear(img, affine_grid(theta, H, W))


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float = 0.25) -> Tuple[Dict[str, Tensor], Tensor]:
    """torch.nn.utils.clip_grad_norm_ (global L2) -- interfaces/super_resolution.py:1083-1084."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-3,
              betas=(0.5, 0.999), eps: float = 1e-8) -> Tuple[Tensor, Tensor, Tensor]:
    """torch.optim.Adam update (no weight decay / amsgrad) -- interfaces/base.py:527, yaml :26-29."""
    m = betas[0] * m + (1 - betas[0]) * g
    v = betas[1] * v + (1 - betas[1]) * g * g
    mhat = m / (1 - betas[0] ** step)
    vhat = v / (1 - betas[1] ** step)
    return p - lr * mhat / (torch.sqrt(vhat) + eps), m, v


PARAM_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")
BUFFER_KEYS = ("infoGen.pe.pe", "tps.inverse_kernel", "tps.padding_matrix",
               "tps.target_coordinate_repr", "tps.target_control_points")


def is_param(key: str) -> bool:
    return not key.endswith(PARAM_SUFFIXES) and key not in BUFFER_KEYS


def train_step(sd: SD, x: Tensor, text_emb: Optional[Tensor], hr: Tensor, *, tatt: bool = True,
               stn: bool = True, drop_on: bool = False, opt_state: Optional[dict] = None,
               step: int = 1, lr: float = 1e-3, tbsrn: bool = False):
    """One reference training step for fixed inputs: forward (train mode), ImageLoss.mean()*100
    (interfaces/super_resolution.py:889-894), backward, clip 0.25, Adam(1e-3,(0.5,0.999)).

    Returns (loss, grads{name: grad or None}, new_sd, opt_state, forward-dict)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_param(k)}
    full = dict(sd)
    full.update(leaves)
    new_stats: dict = {}
    if tbsrn:
        out = tbsrn_forward(full, x, training=True, stn=stn, drop_on=drop_on, new_stats=new_stats)
    else:
        out = generator_forward(full, x, text_emb, training=True, tatt=tatt, stn=stn, drop_on=drop_on,
                                new_stats=new_stats)
    loss = image_loss(out["sr"], hr).mean() * 100.0
    names = list(leaves)
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gs))
    present = {k: g for k, g in grads.items() if g is not None}
    clipped, total = clip_grad_norm(present)
    opt_state = opt_state if opt_state is not None else {}
    new_sd = dict(sd)
    new_sd.update(new_stats)
    for k, g in clipped.items():
        m, v = opt_state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
        p, m, v = adam_step(sd[k], g, m, v, step, lr)
        new_sd[k] = p
        opt_state[k] = (m, v)
    return loss.detach(), grads, new_sd, opt_state, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, total


def train_step_fp64(sd: SD, x: Tensor, text_emb: Optional[Tensor], hr: Tensor, **kw):
    """`train_step` evaluated in float64 (same graph, same weights and inputs widened exactly): the yardstick that separates
    implementation error from the conditioning of the fp32 computation itself (tests: ||g_hip - g_64|| vs ||g_ref32 - g_64||)."""
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    return train_step(sd64, x.double(), None if text_emb is None else text_emb.double(), hr.double(), **kw)
