#!/usr/bin/env python3
"""Would a split-bf16 matrix-core path for the 3x3 convolutions stay inside the parity budget?  CPU only, on the oracle.

bf16 MFMA runs at 16x the fp32-MFMA rate on gfx950.  Split-bf16: every fp32 operand a = hi + lo with hi = bf16(a), lo = bf16(a - hi)
(16 mantissa bits kept), and  a * b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b  (the dropped lo_a lo_b term is 2^-16 relative) accumulated in
fp32: three bf16 products per fp32 product, ~5x the fp32-MFMA rate.  This probe replaces the oracle's conv2d for the 3x3 convolutions
between 64-multiples of channels (the kernels that would change: the SRB convs, block7, the up-sampler conv, forward and both
gradients through autograd) by exactly that arithmetic and reports what moves: eval SR, training loss, every gradient against the
fp64 gradients -- next to the plain fp32 oracle's own distance from fp64 (the yardstick of tests/test_model_gpu.py)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatt_amd  # noqa: E402
from oracle import tatt_oracle as O  # noqa: E402
from oracle.fixtures import randomize_state_dict, make_inputs  # noqa: E402
from tests.util import STRUCTURAL_ZERO_GRAD  # noqa: E402


def split(t):
    hi = t.bfloat16().float()
    lo = (t - hi).bfloat16().float()
    return hi, lo


class SplitConv(torch.autograd.Function):
    """conv2d whose forward, data gradient and weight gradient are each three bf16 x bf16 products with fp32 accumulation."""

    @staticmethod
    def forward(ctx, x, w, pad, terms):
        ctx.save_for_backward(x, w)
        ctx.pad, ctx.terms = pad, terms
        return SplitConv.prod(lambda a, b: F.conv2d(a, b, None, padding=pad), x, w, terms)

    @staticmethod
    def prod(op, a, b, terms):
        ah, al = split(a)
        bh, bl = split(b)
        out = op(ah, bh) + (op(ah, bl) + op(al, bh))
        if terms == 4:
            out = out + op(al, bl)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        pad, terms = ctx.pad, ctx.terms
        dx = SplitConv.prod(lambda g, ww: torch.nn.grad.conv2d_input(x.shape, ww, g, padding=pad), dy, w, terms)
        dw = SplitConv.prod(lambda xx, g: torch.nn.grad.conv2d_weight(xx, w.shape, g, padding=pad), x, dy, terms)
        return dx, dw, None, None


class Patch:
    def __init__(self, terms):
        self.terms = terms

    def __enter__(self):
        self.saved = O.conv2d
        terms = self.terms

        def conv2d(x, w, b, pad):
            if x.dtype == torch.float32 and w.shape[2] == 3 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
                y = SplitConv.apply(x, w, pad, terms)
                return y if b is None else y + b.view(1, -1, 1, 1)
            return self.saved(x, w, b, pad)
        O.conv2d = conv2d

    def __exit__(self, *exc):
        O.conv2d = self.saved


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    m = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    sd = randomize_state_dict(m.state_dict())
    x, tp, hr = make_inputs(4, seed=3)
    with torch.no_grad():
        sr32 = O.generator_forward(sd, x, tp, training=False)["sr"]
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        sr64 = O.generator_forward(sd64, x.double(), tp.double(), training=False)["sr"]
    print("eval forward, B = 4, 16x64 -> 32x128, SR in [-1, 1]; fp32 oracle vs fp64: max|dSR| = %.3e" % float((sr32.double() - sr64).abs().max()))
    for terms in (3, 4):
        with torch.no_grad(), Patch(terms):
            sr_s = O.generator_forward(sd, x, tp, training=False)["sr"]
        print("  split-bf16 (%d products) 3x3 convs: max|SR - SR_fp32| = %.3e, vs fp64 %.3e" % (
            terms, float((sr_s - sr32).abs().max()), float((sr_s.double() - sr64).abs().max())))

    # Gradients: with the STN off.  (With it on, the bilinear sampler amplifies ANY last-bit change of the 20 control points -- DESIGN.md
    # section 2 -- and the STN head's gradients of two fp32 runs already differ by ~1e-2: the reference's own fp32 gradient is 5e-3 ..
    # 8e-3 from fp64 there.  That conditioning noise would drown what this probe measures.)
    def grads(patch, dtype=torch.float32):
        p = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(O.is_param(k)) for k, v in sd.items()}
        if patch:
            patch.__enter__()
        try:
            out = O.generator_forward(p, x.to(dtype), tp.to(dtype), training=True, drop_on=False, stn=False)
            loss = O.image_loss(out["sr"], hr.to(dtype)).mean() * 100.0
            loss.backward()
        finally:
            if patch:
                patch.__exit__()
        return float(loss.detach()), {k: v.grad.double() for k, v in p.items() if v.grad is not None}
    l64, g64 = grads(None, torch.float64)
    l32, g32 = grads(None)
    print("train step (dropout off, STN off): loss fp64 %.7f, fp32 %.7f" % (l64, l32))
    scale = max(float(g.abs().max()) for g in g64.values())
    for terms in (3, 4):
        ls, gs = grads(Patch(terms))
        rows = []
        for k, d in g64.items():
            if STRUCTURAL_ZERO_GRAD.search(k):
                continue
            den = float(d.norm()) + 1e-7 * scale * d.numel() ** 0.5
            e32, es = float((g32[k] - d).norm()) / den, float((gs[k] - d).norm()) / den
            rows.append((es / (3.0 * e32 + 5e-4), k, es, e32))
        rows.sort(reverse=True)
        print("  split-bf16 (%d products): loss %.7f (rel to fp64 %.2e; fp32: %.2e); gradients vs fp64, worst ratio to the test limit 3*e32 + 5e-4:" % (
            terms, ls, abs(ls - l64) / l64, abs(l32 - l64) / l64))
        for r in rows[:6]:
            print("     ratio %.2f  %-56s split %.2e  fp32 %.2e" % r)
        med = sorted(r[2] for r in rows)[len(rows) // 2]
        print("     median split error %.2e (fp32: %.2e)" % (med, sorted(r[3] for r in rows)[len(rows) // 2]))


if __name__ == "__main__":
    main()
