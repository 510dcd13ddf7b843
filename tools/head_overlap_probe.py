#!/usr/bin/env python
"""Do the two latency-bound chains of the forward's head overlap?  hipGraph replays of (a) the query embedding (persistent query-GRU
chain) alone, (b) the STN head alone, (c) both on two streams, (d) the query embedding beside a plain streaming kernel, at B = 48."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tatt_amd
from tatt_amd import functional as Fh, ops
import tatt_amd.tsrn as T

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32).to(dev).train()
x = torch.rand(48, 4, 16, 64, device=dev)
big = torch.randn(48, 16, 64, 64, device=dev)
Fh.sticky_word(dev)
side = torch.cuda.Stream()


def q():
    with torch.no_grad():
        return T._query_pos(m.infoGen, 48, 16, 64)


def s():
    with torch.no_grad():
        return T._stn_forward(x, m.stn_head, False)


def stream_kernel():
    for _ in range(8):
        ops.axpby(big, big, 1.0, 0.5) if hasattr(ops, "axpby") else big.mul_(1.0001)


def both(a, b):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ra = a()
    rb = b()
    main.wait_stream(side)
    return ra, rb


def timeit(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            keep = [fn() for _ in range(5)]
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print("%-60s %7.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
    Fh.sync_check()


timeit("query embedding (persistent chain) alone", q)
timeit("STN head forward alone", s)
timeit("8 streaming launches (25 MB each) alone", stream_kernel)
timeit("query embedding || STN head", lambda: both(q, s))
timeit("query embedding || streaming launches", lambda: both(q, stream_kernel))
timeit("STN head || streaming launches", lambda: both(s, stream_kernel))
