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
    tb2 = tf2 = tprep = float("nan")
    if ops.tplayer2_geom(B, L, S)[0]:
        pk = ops.tplayer2_prep(lp, K, V)
        fargs = (x, qpos, pk, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, not fin, fin, S)
        tf2 = timed(lambda: ops.tplayer2_fwd(*fargs))
        hm = ops.tplayer2_fwd(*fargs)[3]
        args = (x, qpos, K, V, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, None if fin else up, up if fin else None, None, None, True)
        args2 = (x, qpos, pk, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, None if fin else up, up if fin else None, None, None, True, S)
        bwd2 = lambda: ops.tplayer2_bwd(*args2, hmask=hm)
        tb2 = timed(bwd2)
        tprep = timed(lambda: ops.tplayer2_prep(lp, K, V))
        # forward parity: second generation (split bf16) against the first (exact fp32)
        o1 = ops.tplayer_fwd(x, qpos, K, V, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, not fin, fin)
        o2 = ops.tplayer2_fwd(*fargs)
        torch.cuda.synchronize()
        fe = ["%s %.1e" % (n, float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)) for n, a, b in zip(("xout", "fin", "wavg"), o2, o1) if a is not None]
        print("   gen2 forward vs gen1 (rel-max err): " + ", ".join(fe))
        # kernel-level parity: second generation (split bf16) against the first (exact fp32), every output after its reducer
        dx1, dq1, kv1, pp1 = ops.tplayer_bwd(*args)
        dK1, dV1 = ops.tplayer_reduce_kv(kv1, B, L, S)
        dx2, dq2, kv2, fl2, pp2, G2 = ops.tplayer2_bwd(*args2, hmask=hm)
        dK2, dV2 = ops.tplayer2_reduce_kv(kv2, fl2, B, L, S)
        names = ["in_w", "in_b", "out_w", "out_b", "w1", "b1", "w2", "b2", "lnA_w", "lnA_b", "lnB_w", "lnB_b", "lnF_w", "lnF_b"]
        shp = [(64, 64), (64,), (64, 64), (64,), (64, 64), (64,), (64, 64), (64,), (64,), (64,), (64,), (64,), (64,), (64,)]
        g1 = [torch.zeros(*sh, device=dev) for sh in shp]
        g2 = [torch.zeros(*sh, device=dev) for sh in shp]
        if not fin:
            g1[12] = g1[13] = g2[12] = g2[13] = None
        ops.tplayer_reduce_params(pp1, B, L, g1)
        ops.tplayer_reduce_params_g(pp2, G2, g2)
        torch.cuda.synchronize()
        worst = []
        for n, a, b in [("dx", dx2, dx1), ("dqpos", dq2, dq1), ("dK", dK2, dK1), ("dV", dV2, dV1)] + [(n, a, b) for n, a, b in zip(names, g2, g1) if a is not None]:
            e = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)
            worst.append((e, n))
        worst.sort(reverse=True)
        print("   gen2 vs gen1 (rel-max err): " + ", ".join("%s %.1e" % (n, e) for e, n in worst[:6]) + ("   ALL <= 1e-4" if worst[0][0] <= 1e-4 else "   MISMATCH"))
    tok = B * L
    f_fwd = tok * (4 * 2 * 64 * 64 + 2 * 2 * S * 64)                 # four 64x64 products + QK^T + PV per token
    f_bwd = tok * (12 * 2 * 64 * 64 + 6 * 2 * S * 64)                # recompute + data gradients + weight gradients
    b_fwd = tok * 64 * 4 * 3 + (tok * S * 4 if fin else 0)           # x, qpos in; one map out (+ weights map)
    b_bwd = tok * 64 * 4 * 5                                         # x, qpos, upstream in; dx, dqpos out
    print("tplayer B=%d L=%d S=%d p=%.1f fin=%d: fwd %7.1f us %6.1f TFLOP/s %6.0f GB/s | bwd %7.1f us %6.1f TFLOP/s %6.0f GB/s | gen2: fwd %7.1f us, bwd %7.1f us, prep %5.1f us" % (
        B, L, S, p, fin, tf, f_fwd / tf / 1e6, b_fwd / tf / 1e3, tb, f_bwd / tb / 1e6, b_bwd / tb / 1e3, tf2, tb2, tprep))
    return tf, tb


def main():
    from __graft_entry__ import build
    build()
    dev = torch.device("cuda:0")
    layer_case(dev, 2, 64, 26, 0.0, False)         # one work-group, one round: smallest case the second generation takes
    layer_case(dev, 3, 128, 26, 0.1, True)
    for p in (0.0, 0.1):
        layer_case(dev, 48, 1024, 26, p, False)
        layer_case(dev, 48, 1024, 26, p, True)
    layer_case(dev, 48, 26, 26, 0.1, False)
    layer_case(dev, 16, 4096, 26, 0.1, True)


if __name__ == "__main__":
    main()
