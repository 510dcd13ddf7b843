#!/usr/bin/env python
"""Stand-alone times (hipGraph of 20 launches each) of the token kernels of a TBSRN FeatureEnhancer at B = 48 (M = 49,152 tokens, 128
channels): the split-bf16 projections with their epilogues, the token-contraction weight gradient at several split counts, the fp32 GEMM
it replaces."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tatt_amd import ops, functional as Fh

dev = torch.device("cuda:0")
M, N, K = 49152, 128, 128
X, DY, F_, Y = (torch.randn(M, 128, device=dev) for _ in range(4))
W = torch.randn(N, K, device=dev) / 11
b = torch.randn(N, device=dev)
Wpk = torch.empty(N * K, device=dev)
ops.call("tatt_tokgemm_pack", ops.P(W), ops.P(Wpk), N, K, K, 0, ops.stream())
seed = Fh.seed_tensor(dev)
dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)


def timeit(name, fn, nbytes):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 100 * 1e6
    print("%-46s %7.1f us   %6.2f TB/s" % (name, us, nbytes / us / 1e6))


MB = M * 128 * 4
timeit("tokgemm 128x128 plain", lambda: Fh._tokgemm_ex(X, Wpk, b, N, K, out=Y), 2 * MB)
timeit("tokgemm 128x128 accumulate", lambda: Fh._tokgemm_ex(X, Wpk, None, N, K, out=Y, accum=True), 3 * MB)
timeit("tokgemm 128x128 relu + dropout (ffn forward)", lambda: ops.call("tatt_tokgemm_sb_ffn", ops.P(X), ops.P(Wpk), ops.P(b), ops.P(Y), M, N, K, 1, 0.1,
                                                                         ops.P(seed), 7, None, 1.0, ops.stream()), 2 * MB)
timeit("tokgemm 128x128 gated (ffn backward)", lambda: ops.call("tatt_tokgemm_sb_ffn", ops.P(DY), ops.P(Wpk), None, ops.P(Y), M, N, K, 0, 0.0, None, 0,
                                                                 ops.P(F_), 1.0 / 0.9, ops.stream()), 3 * MB)
for S in (64, 128, 256, 512):
    ops.TOK_WGRAD_SPLITS = S
    timeit("tok_wgrad 128x128, %d splits (+ reduce)" % S, lambda: ops.tok_wgrad_sb(DY, X, dW, db), 2 * MB)
ops.TOK_WGRAD_SPLITS = 128
timeit("fp32 GEMM weight gradient (gemm_fast + reduce)", lambda: ops.linear_bwd_weight(DY, X, out=dW, out_ld=K, rowsum=db), 2 * MB)
