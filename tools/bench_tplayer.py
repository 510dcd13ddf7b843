#!/usr/bin/env python3
"""Micro-benchmark of the fused TP-interpreter layer (csrc/tplayer.hip) alone at the benchmark shapes: HIP events on the launch
stream.  Prints kernel time, MFMA FLOP rate and algorithmic HBM rate for forward / backward, decoder / encoder geometry, dropout on / off."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


def layer_case(dev, B, L, S, p, fin):
    from tatt_amd import ops, functional as Fh
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
    x, qpos, K, V = r(B, L, 64), r(B, L, 64), r(B, S, 64), r(B, S, 64)
    lp = (r(192, 64), r(192), r(64, 64), r(64), r(64, 64), r(64), r(64, 64), r(64), r(64) + 1, r(64), r(64) + 1, r(64))
    lnF = (r(64) + 1, r(64)) if fin else None
    seed = Fh.seed_tensor(dev)
    up = r(B, L, 64)
    fwd = lambda: ops.tplayer_fwd(x, qpos, K, V, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, not fin, fin)
    nkv, npp = ops.tplayer_geom(B, L)[2:]
    bwd = lambda: ops.tplayer_bwd(x, qpos, K, V, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, None if fin else up, up if fin else None,
                                  None, None, True)
    tf, tb = timed(fwd), timed(bwd)
    tok = B * L
    f_fwd = tok * (4 * 2 * 64 * 64 + 2 * 2 * S * 64)                 # four 64x64 products + QK^T + PV per token
    f_bwd = tok * (12 * 2 * 64 * 64 + 6 * 2 * S * 64)                # recompute + data gradients + weight gradients
    b_fwd = tok * 64 * 4 * 3 + (tok * S * 4 if fin else 0)           # x, qpos in; one map out (+ weights map)
    b_bwd = tok * 64 * 4 * 5                                         # x, qpos, upstream in; dx, dqpos out
    print("tplayer B=%d L=%d S=%d p=%.1f fin=%d: fwd %7.1f us %6.1f TFLOP/s %6.0f GB/s | bwd %7.1f us %6.1f TFLOP/s %6.0f GB/s" % (
        B, L, S, p, fin, tf, f_fwd / tf / 1e6, b_fwd / tf / 1e3, tb, f_bwd / tb / 1e6, b_bwd / tb / 1e3))
    return tf, tb


def main():
    from __graft_entry__ import build
    build()
    dev = torch.device("cuda:0")
    for p in (0.0, 0.1):
        layer_case(dev, 48, 1024, 26, p, False)
        layer_case(dev, 48, 1024, 26, p, True)
    layer_case(dev, 48, 26, 26, 0.1, False)
    layer_case(dev, 16, 4096, 26, 0.1, True)


if __name__ == "__main__":
    main()
