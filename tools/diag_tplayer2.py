#!/usr/bin/env python3
"""Diagnostic: second-generation TP-layer backward against the first, per 16-token tile (which tiles differ, is it deterministic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case(dev, B, L, S=26, p=0.0, fin=False):
    from tatt_amd import ops, functional as Fh
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
    x, qpos, K, V = r(B, L, 64), r(B, L, 64), r(B, S, 64), r(B, S, 64)
    lp = (r(192, 64), r(192), r(64, 64), r(64), r(64, 64), r(64), r(64, 64), r(64), r(64) + 1, r(64), r(64) + 1, r(64))
    lnF = (r(64) + 1, r(64)) if fin else None
    seed = Fh.seed_tensor(dev)
    up = r(B, L, 64)
    args = (x, qpos, K, V, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, None if fin else up, up if fin else None, None, None, True)
    dx1 = ops.tplayer_bwd(*args)[0]
    pk = ops.tplayer2_prep(lp, K, V)
    hm = ops.tplayer2_fwd(x, qpos, pk, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, not fin, fin, S)[3]
    args2 = (x, qpos, pk, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, None if fin else up, up if fin else None, None, None, True, S)
    outs = [ops.tplayer2_bwd(*args2, hmask=hm)[0] for _ in range(3)]
    torch.cuda.synchronize()
    taken, G = ops.tplayer2_geom(B, L, S)[:2]
    scale = float(dx1.abs().max())
    e = ((outs[0] - dx1).abs().amax(dim=2) / scale).reshape(-1, 16).amax(dim=1)      # per tile
    bad = torch.nonzero(e > 1e-4).flatten().tolist()
    rep = [float((o - outs[0]).abs().max()) for o in outs[1:]]
    print("B=%d L=%d p=%.1f fin=%d: G=%d tiles=%d bad tiles=%d  repeat diffs %s" % (B, L, p, fin, G, e.numel(), len(bad), rep))
    if bad:
        print("   first bad tiles:", bad[:40])
        print("   (tile %% 4 histogram)", [sum(1 for t in bad if t % 4 == k) for k in range(4)], " errs", ["%.1e" % float(e[t]) for t in bad[:8]])
        t = bad[0]
        et = ((outs[0] - dx1).abs() / scale).reshape(-1, 16, 64)[t]
        print("   tile %d: bad tokens %s, bad channels (of worst token) %s" % (t, torch.nonzero(et.amax(1) > 1e-4).flatten().tolist(),
              torch.nonzero(et[et.amax(1).argmax()] > 1e-4).flatten().tolist()))


def main():
    from __graft_entry__ import build
    build()
    dev = torch.device("cuda:0")
    for B, L in [(2, 64), (1, 256), (4, 256), (1, 512), (1, 1024), (2, 1024), (8, 1024), (48, 1024)]:
        case(dev, B, L)


if __name__ == "__main__":
    main()
