#!/usr/bin/env python3
"""GPU-box micro-benchmark of the individual hot kernels at the benchmark shapes (B=48, 16x64 LR), timed with HIP events on
the launch stream.  Prints one line per kernel: average microseconds, achieved TFLOP/s or GB/s.  `--only name` restricts to
one kernel (for rocprofv3 --pmc runs)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd import ops  # noqa: E402
from tatt_amd.build import build  # noqa: E402

build(verbose=False)
ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--match", default="", help="substring filter on the kernel name")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--batch", type=int, default=48)
ap.add_argument("--H", type=int, default=16, help="LR height for the conv3_fwd_sb / tplayer entries (large tile: 32)")
ap.add_argument("--W", type=int, default=64, help="LR width (large tile: 128)")
a = ap.parse_args()
dev = torch.device("cuda:0")
B = a.batch


def timeit(name, fn, flops=0.0, bytes_=0.0, iters=None):
    if (a.only and a.only != name) or (a.match and a.match not in name):
        return
    iters = iters or a.iters
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    extra = ""
    if flops:
        extra += "  %7.1f TFLOP/s" % (flops / us / 1e6)
    if bytes_:
        extra += "  %7.1f GB/s" % (bytes_ / us / 1e3)
    print("%-28s %9.1f us%s" % (name, us, extra), flush=True)


R = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
M = B * 1024

# ---- convolutions ----
x64 = R(B, 16, 64, 64)
w33 = R(64, 64, 3, 3) * 0.05
b64 = R(64)
wp = ops.repack_weight(w33, 0)
y64 = torch.empty_like(x64)
f33 = 2.0 * M * 576 * 64
wt = ops.repack_weight(w33, 2)
timeit("conv3_fwd_t_64_64", lambda: ops.call("tatt_conv3_c64_fwd_t", ops.P(x64), ops.P(wt), ops.P(b64), ops.P(y64), B, 16, 64, 64, 64,
                                              0, 0.0, ops.stream()), f33)
wl16 = ops.repack_weight(w33, 6)
timeit("conv3_fwd_ws16_64_64", lambda: ops.call("tatt_conv3_c64_fwd_ws16", ops.P(x64), ops.P(wl16), ops.P(b64), ops.P(y64), B, 16, 64, 64,
                                                 0, 0.0, ops.stream()), f33)
wsb = ops.repack_weight(w33, 10)
xs_ = R(B, a.H, a.W, 64)
ys_ = torch.empty_like(xs_)
_wsb_cur = ops.repack_weight(w33, ops.LIB.tatt_conv3_sb_packing(B, a.H, a.W, 64, 64, 0, 0))
timeit("conv3_fwd_sb_64_64", lambda: ops.call("tatt_conv3_c64_fwd_sb", ops.P(xs_), 64, 0, ops.P(_wsb_cur), ops.P(b64), ops.P(ys_), B, a.H, a.W, 64, 0, 0.0,
                                               None, None, 0, None, ops.stream()), 2.0 * B * a.H * a.W * 576 * 64, 2.0 * B * a.H * a.W * 64 * 4)
# round 6: both generations of the split-bf16 kernel (tatt_conv3_sb_generation) and its folded variants as the residual blocks run them
from tatt_amd._lib import LIB as _LIB  # noqa: E402
_f3, _b3 = 2.0 * B * a.H * a.W * 576 * 64, 2.0 * B * a.H * a.W * 64 * 4
_sc, _sh = 1.0 + 0.1 * R(64), 0.1 * R(64)
_coef = torch.stack([1.0 + 0.1 * R(64), 0.01 * R(64), 0.01 * R(64)])
_mean, _rstd = 0.1 * R(64), 1.0 + 0.1 * R(64).abs()
_w64 = w33.clone()
_wsb_gen = {1: wsb, 3: ops.repack_weight(w33, 14), 4: ops.repack_weight(w33, 14)}
for _gen in (1, 3, 4):
    def _g(fn, _gen=_gen):
        def run():
            _LIB.tatt_conv3_sb_generation(_gen)
            ops.CONV3_SB_GENERATION = _gen
            try:
                fn()
            finally:
                _LIB.tatt_conv3_sb_generation(4)
                ops.CONV3_SB_GENERATION = 4
        return run
    timeit("conv3_sb_g%d_plain" % _gen, _g(lambda _gen=_gen: ops.call("tatt_conv3_c64_fwd_sb", ops.P(xs_), 64, 0, ops.P(_wsb_gen[_gen]), ops.P(b64), ops.P(ys_), B, a.H, a.W, 64, 0, 0.0,
                                                                None, None, 0, None, ops.stream())), _f3, _b3)
    timeit("conv3_sb_g%d_bn_mish_stats" % _gen, _g(lambda: ops.conv3_bn_forward(xs_, _w64, b64, _sc, _sh, 2, True)), _f3, _b3)
    timeit("conv3_sb_g%d_stats" % _gen, _g(lambda: ops.conv3_bn_forward(xs_, _w64, b64, None, None, 0, True)), _f3, _b3)
    timeit("conv3_sb_g%d_dgrad_in2_epbn" % _gen, _g(lambda: ops.conv3_dgrad_bn(xs_, _w64, ys_, _coef, (xs_, _mean, _rstd, _sc, _sh, 2))), _f3, 2 * _b3)
    timeit("conv3_sb_g%d_dgrad_in2" % _gen, _g(lambda: ops.conv3_dgrad_bn(xs_, _w64, ys_, _coef, None)), _f3, 1.5 * _b3)
timeit("conv3_wgrad_64_64", lambda: ops.conv_wgrad(x64, y64, 64, 3, 3), f33)
for _gen in (1, 2):
    for _G in (64, 128, 256):
        def _wg(_gen=_gen, _G=_G):
            _LIB.tatt_conv3_wgrad_sb_generation(_gen)
            old = ops.CONV3_WGRAD_GROUPS
            ops.CONV3_WGRAD_GROUPS = _G
            try:
                ops.conv_wgrad(x64, y64, 64, 3, 3, want_db=True)
            finally:
                _LIB.tatt_conv3_wgrad_sb_generation(2)
                ops.CONV3_WGRAD_GROUPS = old
        timeit("conv3_wgrad_g%d_G%d" % (_gen, _G), _wg, f33)
ops.CONV3_WGRAD_SB = False
timeit("conv3_wgrad_64_64_fp32", lambda: ops.conv_wgrad(x64, y64, 64, 3, 3), f33)
ops.CONV3_WGRAD_SB = True
xs = x64.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)       # strided view -> generic implicit-GEMM kernel
timeit("conv3_generic_64_64", lambda: ops.conv_fwd(xs, wp, b64, 64, 3, 3, out=y64), f33)
w256 = R(256, 64, 3, 3) * 0.05
wp256 = ops.repack_weight(w256, 0)
y256 = torch.empty(B, 16, 64, 256, device=dev)
wt256 = ops.repack_weight(w256, 2)
timeit("conv3_fwd_t_64_256", lambda: ops.call("tatt_conv3_c64_fwd_t", ops.P(x64), ops.P(wt256), None, ops.P(y256), B, 16, 64, 64, 256,
                                               0, 0.0, ops.stream()), 4 * f33)
wl16_256 = ops.repack_weight(w256, 6)
timeit("conv3_fwd_ws16_64_256", lambda: ops.call("tatt_conv3_c64_fwd_ws16", ops.P(x64), ops.P(wl16_256), None, ops.P(y256), B, 16, 64, 256,
                                                  0, 0.0, ops.stream()), 4 * f33)
wt256d = ops.repack_weight(w256, 3)
timeit("conv3_dgrad_t_256_64", lambda: ops.call("tatt_conv3_c64_fwd_t", ops.P(y256), ops.P(wt256d), None, ops.P(y64), B, 16, 64, 256, 64,
                                                 0, 0.0, ops.stream()), 4 * f33)
timeit("conv3_wgrad_64_256", lambda: ops.conv_wgrad(x64, y256, 256, 3, 3), 4 * f33)
ops.CONV3_WGRAD_SB = False
timeit("conv3_wgrad_64_256_fp32", lambda: ops.conv_wgrad(x64, y256, 256, 3, 3), 4 * f33)
ops.CONV3_WGRAD_SB = True
xhr = R(B, 32, 128, 64)
w99 = R(4, 64, 9, 9) * 0.02
wp99 = ops.repack_weight(w99, 0)
yhr = torch.empty(B, 32, 128, 4, device=dev)
f99 = 2.0 * B * 4096 * 5184 * 4
timeit("conv9_fwd_64_4_hr", lambda: ops.conv_fwd(xhr, wp99, None, 4, 9, 9, out=yhr), f99)
timeit("conv9_fwd_mfma_64_4_hr", lambda: ops.conv2d_forward(xhr, w99, None), f99)
ylr4 = R(B, 16, 64, 64)
w14 = R(64, 4, 9, 9) * 0.02
timeit("conv9_dgrad_mfma_64_4_lr", lambda: ops.conv2d_dgrad(ylr4, w14), f99 / 4)
timeit("conv9_wgrad_64_4_hr", lambda: ops.conv_wgrad(xhr, yhr, 4, 9, 9), f99)
wd99 = ops.repack_weight(w99, 1)       # [81][4][64]: dgrad 4 -> 64 (generic kernel)
timeit("conv9_dgrad_4_64_hr", lambda: ops.conv_fwd(yhr, wd99, None, 64, 9, 9, out=xhr), f99)
x4 = R(B, 16, 64, 4)
w1 = R(64, 4, 9, 9) * 0.05
wp1 = ops.repack_weight(w1, 0)
timeit("conv9_fwd_4_64_lr", lambda: ops.conv_fwd(x4, wp1, b64, 64, 9, 9, out=y64), 2.0 * M * 324 * 64)
timeit("conv9_wgrad_4_64_lr", lambda: ops.conv_wgrad(x4, y64, 64, 9, 9), 2.0 * M * 324 * 64)

# ---- token GEMMs (M = B*1024 tokens) ----
t64, t128, t192 = R(M, 64), R(M, 64), R(M, 192)
W64, W192, W128 = R(64, 64), R(192, 64), R(64, 128)
o64, o192 = torch.empty(M, 64, device=dev), torch.empty(M, 192, device=dev)
timeit("linear_fwd_64_64", lambda: ops.linear_fwd(t64, W64, b64, out=o64), 2.0 * M * 64 * 64, M * 128 * 4)
timeit("linear_fwd_64_96x2", lambda: ops.linear_fwd(t64, W192, None, out=o192), 2.0 * M * 64 * 192, M * 256 * 4)
timeit("linear_fwd_cat128_64", lambda: ops.linear_fwd(t64, W128, b64, x2b=t128, out=o64), 2.0 * M * 128 * 64, M * 192 * 4)
timeit("linear_bwd_input_192_64", lambda: ops.linear_bwd_input(t192, W192, out=o64), 2.0 * M * 192 * 64, M * 256 * 4)
timeit("linear_bwd_weight_64x64", lambda: ops.linear_bwd_weight(t64, t128), 2.0 * M * 64 * 64, M * 128 * 4)
timeit("linear_bwd_weight_192x64", lambda: ops.linear_bwd_weight(t192, t64), 2.0 * M * 192 * 64, M * 256 * 4)
timeit("colsum_64", lambda: ops.colsum(t64), 0, M * 64 * 4)
timeit("colsum_192", lambda: ops.colsum(t192), 0, M * 192 * 4)

# ---- norms / element-wise ----
g64 = torch.ones(64, device=dev)
rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
timeit("bn_stats", lambda: ops.bn_stats(t64, 1e-5, 0.1, rm, rv), 0, M * 64 * 4)
mean, rstd = ops.bn_stats(t64, 1e-5, 0.1, rm, rv)
timeit("bn_apply_mish", lambda: ops.bn_apply(t64, mean, rstd, g64, b64, 2, out=o64), 0, M * 128 * 4)
timeit("bn_bwd_mish", lambda: ops.bn_bwd(t64, t128, mean, rstd, g64, b64, 2, True), 0, M * 64 * 4 * 5)
timeit("ln_fwd", lambda: ops.ln_fwd(t64, t128, g64, b64), 0, M * 64 * 4 * 3)
_, st = ops.ln_fwd(t64, t128, g64, b64)
timeit("ln_bwd", lambda: ops.ln_bwd(t64, t128, o64, st, g64), 0, M * 64 * 4 * 4)

# ---- GRU recurrences: first generation (32-lane group per (sequence, direction)) vs second (one wave each), with / without fragments ----
gi = R(M, 192)
whh, bhh = R(96, 32) * 0.2, R(96) * 0.1
for vert in (True, False):
    geom = ops.seq_geom(B, 16, 64, vert)
    nm = "v" if vert else "h"
    for v2 in (False, True):
        ops.GRU32_V2 = v2
        sfx = nm + ("_v2" if v2 else "")
        timeit("gru32_fwd_" + sfx, lambda: ops.gru32_fwd(gi, whh, bhh, whh, bhh, geom, save=True), 0, M * (192 + 64 + 256) * 4)
        out, gates = ops.gru32_fwd(gi, whh, bhh, whh, bhh, geom, save=True)
        timeit("gru32_bwd_" + sfx, lambda: ops.gru32_bwd(gates, out, t64, whh, whh, geom), 0, M * (256 + 64 * 2 + 192 * 2 + 64) * 4)
    timeit("gru32_bwd_%s_v2_frag" % nm, lambda: ops.gru32_bwd_frag(gates, out, t64, whh, whh, geom), 0, M * (256 + 64 * 2 + 192 + 320) * 4)

# ---- attention core ----
seed = torch.zeros(1, dtype=torch.int64, device=dev)
Q, K, V = R(B, 1024, 64), R(B, 26, 64), R(B, 26, 64)
timeit("attn_fwd", lambda: ops.attn_fwd(Q, K, V, 0.1, seed, 1), 2.0 * B * 1024 * 26 * 64 * 2, M * 64 * 4 * 2)
timeit("attn_bwd", lambda: ops.attn_bwd(Q, K, V, Q, None, 0.1, seed, 1), 2.0 * B * 1024 * 26 * 64 * 5, M * 64 * 4 * 3)

# ---- fused TP-interpreter layer (csrc/tplayer.hip) ----
from tatt_amd import functional as Fh  # noqa: E402
L = a.H * a.W
tx, tq, tK, tV, tup = R(B, L, 64) * 0.3, R(B, L, 64) * 0.3, R(B, 26, 64) * 0.3, R(B, 26, 64) * 0.3, R(B, L, 64) * 0.3
lp = tuple(R(*sh) * 0.3 for sh in ((192, 64), (192,), (64, 64), (64,), (64, 64), (64,), (64, 64), (64,), (64,), (64,), (64,), (64,)))
lnF = (R(64) * 0.3 + 1, R(64) * 0.3)
sd = Fh.seed_tensor(dev)
timeit("tplayer_fwd", lambda: ops.tplayer_fwd(tx, tq, tK, tV, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, sd, 10, 1e-5, False, True),
       B * L * 4 * 2 * 64 * 64, B * L * (64 * 4 * 3 + 26 * 4))
timeit("tplayer_bwd", lambda: ops.tplayer_bwd(tx, tq, tK, tV, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, sd, 10, 1e-5, None, tup, None, None, True),
       B * L * (12 * 2 * 64 * 64 + 2 * 2 * 26 * 64), B * L * 64 * 4 * 5)

# second generation of the backward (csrc/tplayer2.hip): prep (operand packing) + kernel, the relu bits from the forward launch
if ops.tplayer2_geom(B, L, 26)[0]:
    tpk = ops.tplayer2_prep(lp, tK, tV)
    t2f = lambda: ops.tplayer2_fwd(tx, tq, tpk, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, sd, 10, 1e-5, False, True, 26)
    thm = t2f()[3]
    timeit("tplayer2_fwd", t2f, B * L * (4 * 2 * 64 * 64 + 2 * 2 * 26 * 64), B * L * (64 * 4 * 3 + 26 * 4))
    timeit("tplayer2_bwd", lambda: ops.tplayer2_bwd(tx, tq, tpk, lp, lnF, 0.5, 1, 0.1, 0.1, 0.1, sd, 10, 1e-5, None, tup, None, None, True, 26, hmask=thm),
           B * L * (12 * 2 * 64 * 64 + 8 * 2 * 26 * 64), B * L * 64 * 4 * 5)

# ---- TBSRN score-free self-attention (csrc/sattn.hip): P = 1024 tokens, 4 heads x 32 ----
sQ, sK, sV, sdO = (R(B, 1024, 128) for _ in range(4))
sO, slse, sws = torch.empty_like(sQ), torch.empty(B, 4, 1024, device=dev), torch.empty(B, 4, 1024, device=dev)
sdQ, sdK, sdV = torch.empty_like(sQ), torch.empty_like(sQ), torch.empty_like(sQ)
ssc = 32 ** -0.5


def _sattn():
    ops.call("tatt_sattn_fwd", ops.P(sQ), ops.P(sK), ops.P(sV), ops.P(sO), ops.P(slse), B, 1024, 4, ssc, 0.1, ops.P(sd), 100, ops.stream())
    ops.call("tatt_sattn_bwd", ops.P(sQ), ops.P(sK), ops.P(sV), ops.P(sO), ops.P(slse), ops.P(sdO), ops.P(sdQ), ops.P(sdK), ops.P(sdV),
             ops.P(sws), B, 1024, 4, ssc, 0.1, ops.P(sd), 100, ops.stream())


timeit("sattn_fwd_bwd", _sattn, 18.0 * B * 4 * 1024 * 1024 * 32, B * 1024 * 128 * 4 * 11)

# ---- GruBlock weight gradients: three fp32-MFMA GEMMs vs the fused split-bf16 pass (csrc/gruwgrad.hip) ----
gdgi, gdgh, gx, gxb, ghp = R(M, 192), R(M, 192), R(M, 64), R(M, 64), R(M, 64)
gWp, gWhh, gbp, gbhh = (torch.empty(192, 128, device=dev), torch.empty(192, 64, device=dev), torch.empty(192, device=dev),
                        torch.empty(192, device=dev))


def _gru_wgrad_gemms():
    ops.linear_bwd_weight(gdgi, gx, out=gWp, out_ld=128, rowsum=gbp)
    ops.linear_bwd_weight(gdgi, gxb, out=gWp.reshape(-1)[64:], out_ld=128)
    ops.linear_bwd_weight(gdgh, ghp, out=gWhh, rowsum=gbhh)


timeit("gru_wgrad_3gemm", _gru_wgrad_gemms, 2.0 * M * 192 * 192, M * 576 * 4)
for G_ in (64, 128, 256):
    ops.GRU_WGRAD_GROUPS = G_
    timeit("gru_wgrad_sb_G%d" % G_, lambda: ops.gru_wgrad_sb(gdgi, gdgh, gx, gxb, ghp, gWp, gWhh, gbp, gbhh), 2.0 * M * 192 * 192,
           M * 576 * 4)
ops.GRU_WGRAD_GROUPS = 128
# the same gradients from the recurrence's fragment stream (round 4): dgi | gn | hprev fragments 1.25 KB + x | xb 0.25 / 0.5 KB per token
gWhc, gWp64 = torch.empty(192, 32, device=dev), torch.empty(192, 64, device=dev)
for vert in (True, False):
    geom = ops.seq_geom(B, 16, 64, vert)
    out_, gates_ = ops.gru32_fwd(gi, whh, bhh, whh, bhh, geom, save=True)
    _, gfrag = ops.gru32_bwd_frag(gates_, out_, t64, whh, whh, geom)
    for G_ in (64, 128, 256):
        ops.GRU_WGRAD_FRAG_GROUPS = G_
        timeit("gru_wgrad_frag_%s_G%d" % ("v" if vert else "h", G_), lambda: ops.gru_wgrad_frag(gfrag, gx, gxb, geom, gWp, gWhc, gbp, gbhh),
               2.0 * M * 192 * 160, M * (320 + 128) * 4)
    ops.GRU_WGRAD_FRAG_GROUPS = 128
    timeit("gru_wgrad_frag_%s_k64" % ("v" if vert else "h"), lambda: ops.gru_wgrad_frag(gfrag, gx, None, geom, gWp64, gWhc, gbp, gbhh),
           2.0 * M * 192 * 96, M * (320 + 64) * 4)

# ---- split-bf16 token projections (csrc/tokgemm.hip) ----
for (N_, K_, K1_, N1_) in ((192, 128, 64, 192), (192, 64, 64, 192), (128, 192, 192, 64), (64, 192, 192, 64)):
    Xa = R(M, K1_)
    Xb = R(M, K_ - K1_) if K1_ < K_ else None
    Wt = R(N_, K_) * 0.1
    Wk = torch.empty(N_ * K_, device=dev)
    ops.call("tatt_tokgemm_pack", ops.P(Wt), ops.P(Wk), N_, K_, K_, 0, ops.stream())
    timeit("tokgemm_sb_%dx%d" % (N_, K_), lambda: Fh._tokgemm(Xa, Xb, Wk, None, N_, K_, N1_), 2.0 * M * N_ * K_, M * (N_ + K_) * 4)

# ---- query GRU (hidden 512, time axis = the 48 samples): the whole forward / backward chains as the model issues them -----------
qg = torch.nn.GRU(1024, 512, bidirectional=True, batch_first=True).to(dev)
qemb = R(16 * 64, 64) * 0.3                                      # init_factor: (H * W, C)
qnames = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse",
          "bias_hh_l0_reverse"]
qparams = [getattr(qg, n).detach().requires_grad_(True) for n in qnames]
qe = qemb.clone().requires_grad_(True)


def _qgru_fwd():
    with torch.no_grad():
        return Fh.QueryGruFn.apply(qemb, *[p_.detach() for p_ in qparams], B, 16, 64)


timeit("qgru_fwd_chain", _qgru_fwd, 2.0 * B * 64 * 1536 * 512 * 2, 0)
qout = Fh.QueryGruFn.apply(qe, *qparams, B, 16, 64)
qdq = torch.randn_like(qout)


def _qgru_bwd():
    torch.autograd.grad(qout, [qe] + qparams, qdq, retain_graph=True)


timeit("qgru_bwd_chain", _qgru_bwd, 3 * 2.0 * B * 64 * 1536 * 512 * 2, 0)


def timeit_graph(name, fn, iters=20):
    """the same work captured once and replayed as a hipGraph (what the training step does: no host launch cost between the kernels)"""
    if (a.only and a.only != name) or (a.match and a.match not in name):
        return
    st_ = torch.cuda.Stream()
    with torch.cuda.stream(st_):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            fn()
    torch.cuda.synchronize()
    gph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-28s %9.1f us  (hipGraph replay)" % (name, e0.elapsed_time(e1) / iters * 1e3), flush=True)


timeit_graph("qgru_fwd_chain_graph", _qgru_fwd)


# ---- the recurrences alone: 48 / 47 per-step launches (as a hipGraph: no host cost between them) vs ONE persistent launch -----------
def _with(fwd, bwd, fn):
    def run():
        old = Fh.QGRU_CHAIN_FWD, Fh.QGRU_CHAIN_BWD
        Fh.QGRU_CHAIN_FWD, Fh.QGRU_CHAIN_BWD = fwd, bwd
        try:
            return fn()
        finally:
            Fh.QGRU_CHAIN_FWD, Fh.QGRU_CHAIN_BWD = old
    return run


timeit_graph("qgru_fwd_stepwise_graph", _with(False, False, _qgru_fwd))
timeit_graph("qgru_fwd_persistent_graph", _with(True, True, _qgru_fwd))
timeit("qgru_bwd_stepwise_eager", _with(False, False, _qgru_bwd))
timeit("qgru_bwd_persistent_eager", _with(True, True, _qgru_bwd))
_T, _W, _HID = B, 64, 512
_gi = [R(_W, 3 * _HID) * 0.3 for _ in range(2)]
_whh = [R(3 * _HID, _HID) * 0.03 for _ in range(2)]
_whhT = [w_.t().contiguous() for w_ in _whh]
_bhh = [R(3 * _HID) * 0.1 for _ in range(2)]
_hbuf = torch.zeros(2, _T + 1, _W, _HID, device=dev)
_gsave = torch.empty(2, _T, 4, _W, _HID, device=dev)
_sync = torch.zeros(1024, device=dev, dtype=torch.int32)
_dgh = torch.zeros(2, _T, _W, 3 * _HID, device=dev)
_dhseq = R(2, _T, _W, _HID) * 0.1
_dhc = torch.zeros(2, _W, _HID, device=dev)
_dgia = torch.zeros(2, _W, 3 * _HID, device=dev)


def _fwd_chain_only():
    ops.call("tatt_qgru_fwd_chain", ops.P(_gi[0]), ops.P(_gi[1]), ops.P(_whh[0]), ops.P(_whh[1]), ops.P(_bhh[0]), ops.P(_bhh[1]),
             ops.P(_hbuf[0]), ops.P(_hbuf[1]), ops.P(_gsave[0]), ops.P(_gsave[1]), ops.P(_sync), _T, _W, _HID, 0, _T, None, None, None,
             None, None, 0, None, 0, ops.P(_xf[0]) if _sb else None, ops.P(_xf[1]) if _sb else None, ops.stream())


def _bwd_chain_only():
    ops.call("tatt_qgru_bwd_chain", ops.P(_dgh[0]), ops.P(_dgh[1]), ops.P(_whhT[0]), ops.P(_whhT[1]), ops.P(_dhseq[0]), ops.P(_dhseq[1]),
             ops.P(_gsave[0]), ops.P(_gsave[1]), ops.P(_hbuf[0]), ops.P(_hbuf[1]), ops.P(_dhc[0]), ops.P(_dhc[1]), ops.P(_dgia[0]),
             ops.P(_dgia[1]), ops.P(_sync), _T, _W, _HID, 0, _T - 1, 1, ops.P(_xb[0]) if _sb else None, ops.P(_xb[1]) if _sb else None,
             ops.stream())


_xf = torch.zeros(2, _T + 1, _W, _HID, device=dev)
_xb = torch.zeros(2, _T, _W, 3 * _HID, device=dev)
for _sb in (False, True):
    timeit("qgru_fwd_persistent_launch%s" % ("_sb" if _sb else ""), _fwd_chain_only)           # / 48 = per time step
    timeit("qgru_bwd_persistent_launch%s" % ("_sb" if _sb else ""), _bwd_chain_only)           # / 47
print("   persistent launches: error word %d" % int(_sync[1023].item()), flush=True)
# (the backward chain allocates its split-K workspaces through torch outside a Trainer: not capturable on its own)

# ---- query GRU recurrent weight gradient: two fp32-pipe GEMMs vs one split-bf16 launch for both directions (csrc/gruwgrad.hip) ----
qA = [R(B * 64, 1536) for _ in range(2)]
qB = [R(B * 64, 512) for _ in range(2)]
qf = 2 * 2.0 * B * 64 * 1536 * 512


def _qgru_wgrad_gemms():
    for d in range(2):
        ops.linear_bwd_weight(qA[d], qB[d], rowsum=torch.empty(1536, device=dev))


timeit("qgru_wgrad_gemms", _qgru_wgrad_gemms, qf)
if ops.qgru_wgrad_takes(qA[0], qB[0]):
    for sp in (3, 4, 6, 8, 12):
        def _qgru_wgrad_sb(sp=sp):
            old = ops.QGRU_WGRAD_SPLIT
            ops.QGRU_WGRAD_SPLIT = sp
            try:
                ops.qgru_wgrad_sb(qA[0], qA[1], qB[0], qB[1])
            finally:
                ops.QGRU_WGRAD_SPLIT = old
        timeit("qgru_wgrad_sb split %d" % sp, _qgru_wgrad_sb, qf)

# ---- STN head: data gradient of its second convolution (64 -> 32 channels at 8 x 32: 192 output tiles of the generic kernel) ----
sdy = R(B, 8, 32, 64)
sw = R(64, 32, 3, 3) * 0.05
for tiles, wgs in ((128, 256), (256, 512), (256, 768), (256, 1024)):
    def _stn_dgrad(tiles=tiles, wgs=wgs):
        old = ops.CONV_SPLIT_TILES, ops.CONV_SPLIT_WGS
        ops.CONV_SPLIT_TILES, ops.CONV_SPLIT_WGS = tiles, wgs
        try:
            ops.conv2d_dgrad(sdy, sw)
        finally:
            ops.CONV_SPLIT_TILES, ops.CONV_SPLIT_WGS = old
    timeit("stn_dgrad2 split<%d,%d>" % (tiles, wgs), _stn_dgrad, 2.0 * B * 8 * 32 * 576 * 32)

# ---- STN head: weight gradient of its first convolution (4 -> 32 channels at 16 x 64: ONE output tile, 49,152-pixel contraction) ----
sx0, sdy0 = R(B, 16, 64, 4), R(B, 16, 64, 32)
for cap in (128, 256, 512):
    def _stn_wgrad0(cap=cap):
        old = ops.CONV_WGRAD_SPLIT_CAP
        ops.CONV_WGRAD_SPLIT_CAP = cap
        try:
            ops.conv_wgrad(sx0, sdy0, 32, 3, 3)
        finally:
            ops.CONV_WGRAD_SPLIT_CAP = old
    timeit("stn_wgrad1 split cap %d" % cap, _stn_wgrad0, 2.0 * B * 1024 * 36 * 32)

# ---- 9x9 output convolution 64 -> 4 at HR resolution (B x 32 x 128) and block1's data gradient at LR: fp32 Toeplitz MFMA vs split bf16 ----
ox = R(B, 32, 128, 64)
ow = (R(4, 64, 9, 9) * 0.02).requires_grad_(False)
ob = R(4)
dy1 = R(B, 16, 64, 64)
w1 = R(64, 4, 9, 9) * 0.02
f9 = 2.0 * B * 32 * 128 * 64 * 4 * 81
for sbm in (False, True):
    def _c9(sbm=sbm):
        old = ops.CONV9_SB
        ops.CONV9_SB = sbm
        try:
            ops.conv2d_forward(ox, ow, ob)
        finally:
            ops.CONV9_SB = old
    def _c9d(sbm=sbm):
        old = ops.CONV9_SB
        ops.CONV9_SB = sbm
        try:
            ops.conv2d_dgrad(dy1, w1)
        finally:
            ops.CONV9_SB = old
    timeit("conv9_out_fwd_%s" % ("sb" if sbm else "fp32"), _c9, f9)
    timeit("conv9_block1_dgrad_%s" % ("sb" if sbm else "fp32"), _c9d, f9 / 4)

# ---- 9x9 convolutions 4 -> 64: block1 forward (LR) and the output convolution's data gradient (HR): fp32 weight-stationary MFMA vs split bf16 ----
bx = R(B, 16, 64, 4)
bw = R(64, 4, 9, 9) * 0.05
bb = R(64)
ody = R(B, 32, 128, 4)
for sbm in (False, True):
    def _b1(sbm=sbm):
        old = ops.CONV9_SB
        ops.CONV9_SB = sbm
        try:
            ops.conv2d_forward(bx, bw, bb)
        finally:
            ops.CONV9_SB = old
    def _od(sbm=sbm):
        old = ops.CONV9_SB
        ops.CONV9_SB = sbm
        try:
            ops.conv2d_dgrad(ody, ow)
        finally:
            ops.CONV9_SB = old
    timeit("conv9_block1_fwd_%s" % ("sb" if sbm else "fp32"), _b1, f9 / 4)
    timeit("conv9_out_dgrad_%s" % ("sb" if sbm else "fp32"), _od, f9)

# ---- 9x9 weight gradients: output convolution (HR, 64 x 4) and block1 (LR, 4 x 64) ----
gx = R(B, 32, 128, 64)
gdy = R(B, 32, 128, 4)
gx1 = R(B, 16, 64, 4)
gdy1 = R(B, 16, 64, 64)
for sbm in (False, True):
    def _wo(sbm=sbm):
        old = ops.CONV9_SB
        ops.CONV9_SB = sbm
        try:
            ops.conv_wgrad(gx, gdy, 4, 9, 9)
        finally:
            ops.CONV9_SB = old
    def _wb(sbm=sbm):
        old = ops.CONV9_SB
        ops.CONV9_SB = sbm
        try:
            ops.conv_wgrad(gx1, gdy1, 64, 9, 9)
        finally:
            ops.CONV9_SB = old
    timeit("conv9_out_wgrad_%s" % ("sb" if sbm else "fp32"), _wo, f9)
    timeit("conv9_block1_wgrad_%s" % ("sb" if sbm else "fp32"), _wb, f9 / 4)
