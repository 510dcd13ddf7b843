#!/usr/bin/env python
"""A/B of the TBSRN self-attention generations (csrc/sattn.hip exact fp32 vs csrc/sattn2.hip split bf16): results against a float64
torch restatement, masks bit-identical between generations, per-kernel time of a hipGraph of 20 forward + backward pairs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tatt_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def run(gen, Q, K, V, dO, pdrop, sd, bwd=True, keep_bits=False):
    B, Pn, E = Q.shape
    h = E // 32
    ops.LIB.tatt_sattn_generation(int(gen))
    O, lse, ws = torch.empty_like(Q), torch.empty(B, h, Pn, device=dev), torch.empty(B, h, Pn, device=dev)
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    sc = 32 ** -0.5
    bits = torch.empty(B * h * Pn * Pn // 32, device=dev, dtype=torch.int32) if keep_bits else None
    ops.call("tatt_sattn_fwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(bits), B, Pn, h, sc, pdrop, ops.P(sd), 100, ops.stream())
    if bwd:
        ops.call("tatt_sattn_bwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(dO), ops.P(bits), ops.P(dQ), ops.P(dK), ops.P(dV), ops.P(ws),
                 B, Pn, h, sc, pdrop, ops.P(sd), 100, ops.stream())
    torch.cuda.synchronize()
    return O, lse, dQ, dK, dV


def ref64(Q, K, V, dO):
    B, Pn, E = Q.shape
    h = E // 32
    q, k, v = (t.double().view(B, Pn, h, 32).transpose(1, 2).detach().requires_grad_(True) for t in (Q, K, V))
    s = (q @ k.transpose(-1, -2)) * 32 ** -0.5
    p = torch.softmax(s, -1)
    o = (p @ v).transpose(1, 2).reshape(B, Pn, E)
    o.backward(dO.double())
    back = lambda t: t.grad.transpose(1, 2).reshape(B, Pn, E)
    return o.detach(), torch.logsumexp(s, -1).detach(), back(q), back(k), back(v)


def err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


bwd = "--fwd" not in sys.argv
if "--pmc" in sys.argv:            # three plain fwd + bwd pairs of the B = 48 case (for rocprofv3 --pmc passes)
    Q, K, V, dO = (torch.randn(48, 1024, 128, device=dev) for _ in range(4))
    sd = torch.tensor([77], dtype=torch.int64, device=dev)
    for _ in range(3):
        run(2, Q, K, V, dO, 0.1, sd, True, keep_bits=True)
        run(2, Q, K, V, dO, 0.0, sd, True)
    sys.exit(0)
for (B, Pn, h) in ((2, 128, 2), (1, 256, 4), (3, 1024, 4)):
    Q, K, V, dO = (torch.randn(B, Pn, 32 * h, device=dev) * s for s in (1.5, 1.5, 1.0, 1.0))
    sd = torch.tensor([0x1234567, 0], dtype=torch.int32, device=dev).view(torch.int64)
    r = ref64(Q, K, V, dO)
    for gen in (1, 2):
        g = run(gen, Q, K, V, dO, 0.0, sd, bwd)
        print(f"B={B} P={Pn} h={h} gen {gen} vs fp64:", " ".join(f"{n} {err(a, b):.2e}" for n, a, b in zip(("O", "lse", "dQ", "dK", "dV"), g, r)))
    a, b = run(1, Q, K, V, dO, 0.1, sd, bwd), run(2, Q, K, V, dO, 0.1, sd, bwd)
    print(f"  dropout 0.1, gen 2 vs gen 1:", " ".join(f"{n} {err(x, y):.2e}" for n, x, y in zip(("O", "lse", "dQ", "dK", "dV"), b, a)))
    c = run(2, Q, K, V, dO, 0.1, sd, bwd, keep_bits=True)
    print(f"  dropout 0.1, gen 2 with keep bits vs recomputed masks:", " ".join("%s %s" % (n, "identical" if torch.equal(x, y) else "%.2e" % err(x, y)) for n, x, y in zip(("O", "lse", "dQ", "dK", "dV"), c, b)))

B = 48
Q, K, V, dO = (torch.randn(B, 1024, 128, device=dev) for _ in range(4))
sd = torch.tensor([77], dtype=torch.int64, device=dev)
for gen, kbits, pd in ((1, False, 0.1), (2, False, 0.1), (2, True, 0.1), (1, False, 0.0), (2, False, 0.0)):
    run(gen, Q, K, V, dO, pd, sd, bwd)
    bits = torch.empty(B * 4 * 1024 * 1024 // 32, device=dev, dtype=torch.int32) if kbits else None
    g = torch.cuda.CUDAGraph()
    O, lse, ws = torch.empty_like(Q), torch.empty(B, 4, 1024, device=dev), torch.empty(B, 4, 1024, device=dev)
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                ops.call("tatt_sattn_fwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(bits), B, 1024, 4, 32 ** -0.5, pd, ops.P(sd), 100, ops.stream())
                if bwd:
                    ops.call("tatt_sattn_bwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(dO), ops.P(bits), ops.P(dQ), ops.P(dK), ops.P(dV),
                             ops.P(ws), B, 1024, 4, 32 ** -0.5, pd, ops.P(sd), 100, ops.stream())
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print(f"gen {gen}{chr(32) + chr(43) + chr(32) + chr(98) + chr(105) + chr(116) + chr(115) if kbits else str()}: {'fwd+bwd' if bwd else 'fwd'} {(time.perf_counter() - t0) / 50 * 1e6:.1f} us at B = 48, P = 1024, 4 heads, dropout {pd}")


def keep_reference(seed, site, n, pdrop):
    """dropout_keep of csrc/common.h for flat indices 0 .. n-1 (n < 2^32), in int64 arithmetic modulo 2^32"""
    M = 0xFFFFFFFF
    k0 = (seed & M) ^ ((site * 0x9E3779B9) & M)
    k1 = ((seed >> 32) + site * 0x85EBCA77) & M
    h = torch.arange(n, device=dev, dtype=torch.int64) ^ k0
    h = ((h ^ (h >> 16)) * 0x85EBCA6B) & M
    h = (h + k1) & M
    h = ((h ^ (h >> 13)) * 0xC2B2AE35) & M
    h = h ^ (h >> 16)
    return h >= int(pdrop * 4294967296.0)


if "--bits" in sys.argv:
    B, Pn, h = 1, 128, 1
    Q, K, V, dO = (torch.randn(B, Pn, 32 * h, device=dev) for _ in range(4))
    seed = 0x1234567
    sd = torch.tensor([seed], dtype=torch.int64, device=dev)
    ops.LIB.tatt_sattn_generation(2)
    O, lse = torch.empty_like(Q), torch.empty(B, h, Pn, device=dev)
    bits = torch.zeros(B * h * Pn * Pn // 32, device=dev, dtype=torch.int32)
    ops.call("tatt_sattn_fwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(bits), B, Pn, h, 32 ** -0.5, 0.1, ops.P(sd), 100, ops.stream())
    torch.cuda.synchronize()
    keep = keep_reference(seed, 100, B * h * Pn * Pn, 0.1).view(B * h, Pn, Pn)
    nb = Pn // 32
    w = bits.view(B * h, nb, nb, 32).to(torch.int64) & 0xFFFFFFFF
    got = torch.zeros_like(keep)
    for d in range(32):
        v, half = d >> 1, d & 1
        key = (v & 3) + 8 * (v >> 2) + 4 * half
        for qq in range(32):
            got[:, qq::32, key::32] = ((w[:, :, :, d] >> qq) & 1).bool()
    print("keep bits vs the hash: mismatches", int((got != keep).sum()), "of", keep.numel(), " kept fraction", float(got.float().mean()))
    bad = (got != keep).nonzero()
    import collections
    print("by (query % 32):", sorted(collections.Counter((bad[:, 1] % 32).tolist()).items()))
    print("by (key % 32):", sorted(collections.Counter((bad[:, 2] % 32).tolist()).items()))
    print("by (query block, key block):", sorted(collections.Counter(zip((bad[:, 1] // 32).tolist(), (bad[:, 2] // 32).tolist())).items()))
    print("wrong value is 'kept':", int(got[got != keep].sum()), "of", int((got != keep).sum()))
