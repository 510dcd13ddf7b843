#!/usr/bin/env python3
"""GPU time of tatt_conv3_c64_fwd_sb (plain variant) per generation over batch sizes: 100 launches captured into one hipGraph, replayed
(no host gaps).  The slope over B / 16 is the per-tile time of a work-group, the intercept the fixed cost of a launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tatt_amd import ops  # noqa: E402
from tatt_amd._lib import LIB  # noqa: E402

dev = torch.device("cuda:0")


def timeit(run, n=100):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


gens = [int(g) for g in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,3,4".split(","))]
if len(sys.argv) > 2:
    LIB.tatt_conv3_debug_wrep(int(sys.argv[2]))      # (S2_DEBUG builds only)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev)
print("B      " + "".join("  gen %d (us)" % g for g in gens))
for B in (16, 32, 48, 96, 144, 192):
    x = torch.randn(B, 16, 64, 64, device=dev)
    y = torch.empty_like(x)
    row = "%-6d" % B
    for gen in gens:
        LIB.tatt_conv3_sb_generation(gen)
        wl = ops.repack_weight(w, LIB.tatt_conv3_sb_packing(B, 16, 64, 64, 64, 0, 0))
        row += "  %10.2f" % timeit(lambda: ops.call("tatt_conv3_c64_fwd_sb", ops.P(x), 64, 0, ops.P(wl), ops.P(b), ops.P(y), B, 16, 64, 64, 0, 0.0,
                                                     None, None, 0, None, ops.stream()))
    print(row, flush=True)
LIB.tatt_conv3_sb_generation(4)
