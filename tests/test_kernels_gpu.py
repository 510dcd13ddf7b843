"""Per-kernel parity: each HIP operator (through the C ABI, forward and backward) against the CPU oracle /
plain fp32 torch on the same seeded inputs.  Tolerances are fp32 round-off class (rtol 2e-4 / atol 2e-5 on
values, 5e-4 on gradients)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tatt_oracle as O
from tests.util import compare_fn, check_close

pytestmark = pytest.mark.gpu


def R(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(100, 70, 37), (256, 64, 64), (64, 192, 128), (33, 5, 300)])
def test_gemm_nt_bias_act(dev, M, N, K):
    from tatt_amd import ops
    A, W, b = R(M, K), R(N, K, seed=1), R(N, seed=2)
    for act, alpha in ((0, 1.0), (1, 0.5)):
        y = ops.linear_fwd(A.to(dev), W.to(dev), b.to(dev), act=act, alpha=alpha)
        ref = alpha * (A @ W.t() + b)
        ref = torch.relu(ref) if act == 1 else ref
        check_close("gemm_nt", y, ref)


def test_gemm_concat_beta_splitk(dev):
    from tatt_amd import ops
    M = 300
    A1, A2, W, b = R(M, 64), R(M, 64, seed=1), R(64, 128, seed=2), R(64, seed=3)
    y = ops.linear_fwd(A1.to(dev), W.to(dev), b.to(dev), x2b=A2.to(dev))
    check_close("concat", y, torch.cat([A1, A2], 1) @ W.t() + b)
    dy = R(M, 64, seed=4)
    dx = ops.linear_bwd_input(dy.to(dev), W.to(dev), col0=64, ncols=64)
    check_close("bwd_input_cols", dx, dy @ W[:, 64:])
    acc = R(M, 64, seed=5).to(dev)
    ref = acc.cpu() + dy @ W[:, :64]
    ops.linear_bwd_input(dy.to(dev), W.to(dev), col0=0, ncols=64, out=acc, beta=1.0)
    check_close("beta", acc, ref)
    big = R(5000, 96, seed=6), R(5000, 40, seed=7)
    dW = ops.linear_bwd_weight(big[0].to(dev), big[1].to(dev))
    check_close("bwd_weight_splitk", dW, big[0].t() @ big[1], rtol=5e-4, atol=2e-4)


def test_gemm_batched_strided(dev):
    from tatt_amd import ops
    Z, M, N, K = 3, 50, 40, 24
    A, B = R(Z, M, K), R(Z, K, N, seed=1)
    C = torch.zeros(Z, M, N, device=dev)
    ops.gemm(A.to(dev), K, 1, B.to(dev), N, 1, C, N, 1, M, N, K, Z=Z, bsA=M * K, bsB=K * N, bsC=M * N)
    check_close("batched", C, A @ B)


def test_linear_fn_autograd(dev):
    from tatt_amd import functional as Fh
    x, xb, W, b = R(2, 50, 64), R(2, 50, 64, seed=1), R(32, 128, seed=2, scale=0.2), R(32, seed=3)
    compare_fn("linear_cat_relu",
               lambda x, xb, W, b: Fh.linear(x, W, b, act=1, alpha=0.7, xb=xb),
               lambda x, xb, W, b: torch.relu(0.7 * (torch.cat([x, xb], -1) @ W.t() + b)),
               [x, xb, W, b], dev)


def test_gemm_aligned_fast_path(dev):
    """Every operand-orientation combination of the aligned fast path (gemm_fast_kernel) against fp64 torch: linear forward
    (A k-contiguous, W k-contiguous), K-concatenated input, input gradient (W column-contiguous, accumulate), weight gradient
    with split-K and the bias-gradient row sums, batched with strides, activation + alpha."""
    from tatt_amd import ops
    M, K, N = 1024, 128, 192
    x, W, b = R(M, K), R(N, K, seed=1, scale=0.1), R(N, seed=2)
    xd, Wd, bd = x.to(dev), W.to(dev), b.to(dev)
    ref = 0.7 * (x.double() @ W.double().t() + b.double())
    check_close("fast_fwd", ops.linear_fwd(xd, Wd, bd, alpha=0.7), ref.float(), 2e-4, 2e-4)
    check_close("fast_fwd_relu", ops.linear_fwd(xd, Wd, bd, act=1), torch.relu(ref / 0.7).float(), 2e-4, 2e-4)
    y = ops.linear_fwd(xd[:, :64].contiguous(), Wd, bd, x2b=xd[:, 64:].contiguous())
    check_close("fast_concat", y, (ref / 0.7).float(), 2e-4, 2e-4)
    # column-strided views of one buffer (row pitch 128, 64 columns each): still 16-byte aligned rows
    y = ops.linear_fwd(xd[:, :64], Wd, bd, x2b=xd[:, 64:])
    check_close("fast_concat_views", y, (ref / 0.7).float(), 2e-4, 2e-4)
    dy = R(M, N, seed=3)
    dx0 = R(M, K, seed=4)
    dx = dx0.to(dev).clone()
    ops.linear_bwd_input(dy.to(dev), Wd, out=dx, beta=1.0)
    check_close("fast_bwd_input", dx, (dy.double() @ W.double() + dx0.double()).float(), 2e-4, 2e-4)
    dxc = ops.linear_bwd_input(dy.to(dev), Wd, col0=64, ncols=64, alpha=0.5)
    check_close("fast_bwd_input_cols", dxc, (0.5 * dy.double() @ W.double()[:, 64:]).float(), 2e-4, 2e-4)
    Mb = 49152 // 4
    xb_, dyb = R(Mb, 128, seed=5), R(Mb, 192, seed=6)
    db = torch.empty(192, device=dev)
    dW = ops.linear_bwd_weight(dyb.to(dev), xb_.to(dev), rowsum=db)
    check_close("fast_bwd_weight", dW, (dyb.double().t() @ xb_.double()).float(), 5e-4, 5e-3)
    check_close("fast_bias_grad", db, dyb.double().sum(0).float(), 5e-4, 5e-3)
    # ragged row count (48 x 26 text tokens = 1248 = 19.5 tiles): rows past the end are computed but never stored
    xr, Wr, br = R(1248, 64, seed=9), R(192, 64, seed=10, scale=0.1), R(192, seed=11)
    yr = torch.full((1248 + 64, 192), 7.0, device=dev)
    ops.linear_fwd(xr.to(dev), Wr.to(dev), br.to(dev), act=1, out=yr[:1248])
    check_close("fast_ragged_fwd", yr[:1248], torch.relu(xr.double() @ Wr.double().t() + br.double()).float(), 2e-4, 2e-4)
    assert float((yr[1248:] - 7.0).abs().max()) == 0.0
    dxr = ops.linear_bwd_input(yr[:1248].contiguous(), Wr.to(dev))
    check_close("fast_ragged_bwd_input", dxr, (yr[:1248].double().cpu() @ Wr.double()).float(), 2e-4, 2e-4)
    # narrow outputs (N = 32: the per-head products of TBSRN's attention) take the 128 x 32 tile variant: all four operand
    # orientations, batched with strides, ragged M for row-major A
    Z = 2
    A, B = R(Z, 256, 64, seed=12), R(Z, 64, 32, seed=13)
    ref = (A.double() @ B.double()).float()
    for a_t in (False, True):
        for b_t in (False, True):
            Ad = (A.transpose(1, 2).contiguous() if a_t else A).to(dev)
            Bd = (B.transpose(1, 2).contiguous() if b_t else B).to(dev)
            C = torch.zeros(Z, 256, 32, device=dev)
            ops.gemm(Ad, *((1, 256) if a_t else (64, 1)), Bd, *((1, 64) if b_t else (32, 1)), C, 32, 1, 256, 32, 64,
                     Z=Z, bsA=256 * 64, bsB=64 * 32, bsC=256 * 32, alpha=0.5)
            check_close("fast32_%d%d" % (a_t, b_t), C, 0.5 * ref, 2e-4, 2e-4)
    C = torch.full((Z, 200 + 8, 32), 3.0, device=dev)
    ops.gemm(A.to(dev), 64, 1, B.to(dev), 32, 1, C, 32, 1, 200, 32, 64, Z=Z, bsA=256 * 64, bsB=64 * 32, bsC=208 * 32)
    check_close("fast32_ragged", C[:, :200], ref[:, :200], 2e-4, 2e-4)
    assert float((C[:, 200:] - 3.0).abs().max()) == 0.0
    Z = 3
    A, B = R(Z, 128, 64, seed=7), R(Z, 64, 128, seed=8)
    C = torch.zeros(Z, 128, 128, device=dev)
    ops.gemm(A.to(dev), 64, 1, B.to(dev), 128, 1, C, 128, 1, 128, 128, 64, Z=Z, bsA=128 * 64, bsB=64 * 128, bsC=128 * 128)
    check_close("fast_batched", C, (A.double() @ B.double()).float(), 2e-4, 2e-4)
    At = A.transpose(1, 2).contiguous()                     # (Z, 64, 128): A^T stored, i.e. A is row-index-contiguous
    C2 = torch.zeros(Z, 128, 128, device=dev)
    ops.gemm(At.to(dev), 1, 128, B.to(dev), 128, 1, C2, 128, 1, 128, 128, 64, Z=Z, bsA=128 * 64, bsB=64 * 128, bsC=128 * 128)
    check_close("fast_batched_mc", C2, (A.double() @ B.double()).float(), 2e-4, 2e-4)
    Ct = torch.zeros(Z, 128, 128, device=dev)               # column-major C: scalar epilogue
    ops.gemm(A.to(dev), 64, 1, B.to(dev), 128, 1, Ct, 1, 128, 128, 128, 64, Z=Z, bsA=128 * 64, bsB=64 * 128, bsC=128 * 128)
    check_close("fast_batched_ct", Ct.transpose(1, 2), (A.double() @ B.double()).float(), 2e-4, 2e-4)


# ------------------------------------------------------------------------------------------- conv
@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [(2, 16, 64, 64, 64, 3), (1, 16, 64, 64, 256, 3), (2, 16, 64, 4, 64, 9),
                                             (1, 32, 128, 64, 4, 9), (3, 8, 32, 32, 64, 3), (2, 1, 2, 256, 256, 3),
                                             (2, 5, 7, 4, 32, 3)])
@pytest.mark.parametrize("contig", [True, False])
def test_conv2d_fn(dev, B, H, W, Cin, Cout, k, contig):
    """contig=True feeds an NHWC-contiguous map (the layout inside the network: specialised LDS-halo / VALU kernels where the
    shape allows); contig=False feeds an NCHW tensor through strides (generic implicit-GEMM kernel)."""
    from tatt_amd import functional as Fh
    x = R(B, Cin, H, W)
    w = R(Cout, Cin, k, k, seed=1, scale=1.0 / math.sqrt(Cin * k * k))
    b = R(Cout, seed=2)

    def hip(x, w, b):
        xin = x.permute(0, 2, 3, 1)
        if contig:
            xin = xin.contiguous()
        return Fh.conv2d(xin, w, b).permute(0, 3, 1, 2)

    compare_fn("conv%dx%d_%d_%d" % (k, k, Cin, Cout), hip,
               lambda x, w, b: F.conv2d(x, w, b, padding=k // 2), [x, w, b], dev, grtol=1e-3)


def test_conv3_large_tile_and_accumulate(dev):
    """W = 128 (two 64-pixel segments per row), Cin = 128 (two channel chunks), beta accumulate."""
    from tatt_amd import ops
    x, w, b = R(2, 128, 8, 128), R(64, 128, 3, 3, seed=1, scale=0.03), R(64, seed=2)
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    y0 = R(2, 8, 128, 64, seed=3).to(dev)
    y = y0.clone()
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1) + y0.cpu()
    # LDS-staged-filter kernel (filter packed [9][Cout][Cin])
    ops.call("tatt_conv3_c64_fwd_t", ops.P(xin), ops.P(ops.repack_weight(w.to(dev), 2)), ops.P(b.to(dev)), ops.P(y), 2, 8, 128, 128, 64,
             0, 1.0, ops.stream())
    check_close("conv3_t_beta", y, ref, rtol=5e-4, atol=5e-5)
    # generic implicit-GEMM kernel (filter packed [9][Cin][Cout])
    y = y0.clone()
    ops.conv_fwd(xin, ops.repack_weight(w.to(dev), 0), b.to(dev), 64, 3, 3, out=y, beta=1.0)
    check_close("conv3_generic_beta", y, ref, rtol=5e-4, atol=5e-5)


def test_conv_tanh_epilogue(dev):
    from tatt_amd import functional as Fh
    x, w, b = R(1, 16, 8, 8), R(4, 16, 3, 3, seed=1, scale=0.2), R(4, seed=2)
    compare_fn("conv_tanh", lambda x, w, b: Fh.conv2d(x.permute(0, 2, 3, 1), w, b, 3).permute(0, 3, 1, 2),
               lambda x, w, b: torch.tanh(F.conv2d(x, w, b, padding=1)), [x, w, b], dev)


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("C,M", [(64, 2048), (512, 6), (32, 300)])
def test_batchnorm_act(dev, training, act, C, M):
    from tatt_amd import functional as Fh
    x = R(M, C) * 1.5 + 0.3
    g, b = 1 + 0.1 * R(C, seed=1), 0.1 * R(C, seed=2)
    rm, rv = 0.1 * R(C, seed=3), 1 + 0.3 * torch.rand(C)
    bn = torch.nn.BatchNorm1d(C)
    bn.train(training)
    actf = {0: lambda t: t, 1: torch.relu, 2: O.mish}[act]

    def ref(x, g, b):
        sd = {"bn.weight": g, "bn.bias": b, "bn.running_mean": rm, "bn.running_var": rv,
              "bn.num_batches_tracked": torch.tensor(0)}
        ns = {}
        y = actf(O.batch_norm(x, sd, "bn", training, new_stats=ns))
        ref.ns = ns
        return y

    def hip(x, g, b):
        bn_d = torch.nn.BatchNorm1d(C).to(dev)
        bn_d.train(training)
        with torch.no_grad():
            bn_d.running_mean.copy_(rm); bn_d.running_var.copy_(rv)
        bn_d.weight, bn_d.bias = torch.nn.Parameter(g), torch.nn.Parameter(b)
        y = Fh.BatchNormActFn.apply(x, g, b, bn_d.running_mean, bn_d.running_var, training, 0.1, 1e-5, act)
        hip.bn = bn_d
        return y

    compare_fn("bn", hip, ref, [x, g, b], dev, grtol=1e-3)
    if training:
        check_close("running_mean", hip.bn.running_mean, ref.ns["bn.running_mean"])
        check_close("running_var", hip.bn.running_var, ref.ns["bn.running_var"])


def test_layernorm(dev):
    from tatt_amd import functional as Fh
    a, b, g, be = R(3, 100, 64), R(3, 100, 64, seed=1), 1 + 0.1 * R(64, seed=2), 0.1 * R(64, seed=3)
    ln = torch.nn.LayerNorm(64)
    compare_fn("ln_res", lambda a, b, g, be: Fh.LayerNormFn.apply(a, b, g, be, 1e-5, 0, 0.0, 0),
               lambda a, b, g, be: O.layer_norm(a + b, g, be), [a, b, g, be], dev)
    compare_fn("ln", lambda a, g, be: Fh.LayerNormFn.apply(a, None, g, be, 1e-5, 0, 0.0, 0),
               lambda a, g, be: O.layer_norm(a, g, be), [a, g, be], dev)


# ------------------------------------------------------------------------------------------- element-wise
def test_prelu_pixelshuffle_maxpool_tanh(dev):
    from tatt_amd import functional as Fh
    x, al = R(2, 16, 64, 8), torch.tensor([0.25])
    compare_fn("prelu", Fh.prelu, O.prelu, [x, al], dev)
    x = R(2, 4, 6, 32)
    compare_fn("pixel_shuffle_mish", lambda x: Fh.PixelShuffleActFn.apply(x, 2),
               lambda x: O.mish(O.pixel_shuffle2(x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1), [x], dev)
    x = R(2, 8, 16, 12)
    for kh, kw in ((2, 2), (1, 2)):
        compare_fn("maxpool", lambda x: Fh.max_pool(x, kh, kw),
                   lambda x: F.max_pool2d(x.permute(0, 3, 1, 2), (kh, kw)).permute(0, 2, 3, 1), [x], dev)
    compare_fn("tanh", lambda x: Fh.ActFn.apply(x, 3), torch.tanh, [x], dev)
    a, b = R(5, 7, 3, 2), R(5, 7, 3, 2, seed=1)
    compare_fn("add", Fh.add, lambda a, b: a + b, [a, b], dev)
    compare_fn("mean2", Fh.MeanOf2Fn.apply, lambda a, b: 0.5 * (a + b), [a, b], dev)
    compare_fn("permute", lambda a: Fh.Permute4dFn.apply(a, (0, 3, 1, 2)), lambda a: a.permute(0, 3, 1, 2), [a], dev)


def test_add_n_equals_chain_of_adds(dev):
    """tatt_add_n: ((s0 + s1) + s2) + ... in one launch, bit-identical to the chain of binary adds (any size, incl. a ragged tail)."""
    from tatt_amd import ops
    for shape in [(48, 16, 64, 64), (3, 7, 5), (1, 9)]:
        ts = [R(*shape, seed=k).to(dev) for k in range(5)]
        want = ts[0]
        for t in ts[1:]:
            want = want + t
        assert torch.equal(ops.add_n(ts), want)
        assert torch.equal(ops.add_n(ts[:2]), ts[0] + ts[1])


def test_linear_bwd_input_halves(dev):
    """One 2-batch GEMM for both halves of a concatenated input's gradient == the two separate column-range GEMMs."""
    from tatt_amd import ops
    dy, W = R(1024, 192).to(dev), R(192, 128, seed=1).to(dev)
    a, b = ops.linear_bwd_input_halves(dy, W)
    assert a.is_contiguous() and b.is_contiguous()
    assert torch.equal(a, ops.linear_bwd_input(dy, W, col0=0, ncols=64))
    assert torch.equal(b, ops.linear_bwd_input(dy, W, col0=64, ncols=64))
    check_close("halves", torch.cat([a, b], 1), dy.cpu() @ W.cpu(), rtol=1e-4, atol=1e-4)


def test_conv3_wgrad_bias_gradient(dev):
    """The 3x3 64-channel weight-gradient kernel's bias gradient (sum of the dy operands it streams) against dy.sum over pixels."""
    from tatt_amd import ops
    x, dy = R(3, 16, 64, 64).to(dev), R(3, 16, 64, 128, seed=1).to(dev)
    dw, db = ops.conv_wgrad(x, dy, 128, 3, 3, want_db=True)
    # split-bf16 kernel (round 4): every dy operand keeps 16 mantissa bits (hi + lo), i.e. 2^-17 relative per term; the sum of M = 3072
    # unit-scale terms carries ~2^-17 sqrt(M) = 4e-4 of absolute error (measured 3.2e-4) -- 6e-6 of the sum's natural scale sqrt(M)
    check_close("db", db, dy.cpu().double().sum((0, 1, 2)).float(), rtol=1e-5, atol=2e-3)
    ops.CONV3_WGRAD_SB = False
    try:
        _, db32 = ops.conv_wgrad(x, dy, 128, 3, 3, want_db=True)                   # the exact-fp32 kernel: round-off only
    finally:
        ops.CONV3_WGRAD_SB = True
    check_close("db fp32", db32, dy.cpu().double().sum((0, 1, 2)).float(), rtol=1e-5, atol=1e-4)
    assert torch.equal(dw, ops.conv_wgrad(x, dy, 128, 3, 3))


def test_layernorm_fused_dropout_equals_separate(dev):
    """LayerNorm(a + Dropout(b)) in one kernel (tatt_ln_fwd / tatt_ln_bwd with pdrop) == dropout kernel followed by the plain
    LayerNorm, bit for bit: same mask (seed word, site, flat index), same arithmetic; gradients of a, b, gamma, beta included."""
    from tatt_amd import functional as Fh
    Fh.set_seed(dev, 11)
    Fh.begin_training_forward(dev)
    ln = torch.nn.LayerNorm(64).to(dev)
    outs = []
    for fused in (True, False):
        a = R(5, 77, 64).to(dev).requires_grad_()
        b = R(5, 77, 64, seed=1).to(dev).requires_grad_()
        ln.zero_grad()
        if fused:
            y = Fh.layer_norm(a, b, ln, 0.1, True, 9)
        else:
            y = Fh.layer_norm(a, Fh.dropout(b, 0.1, True, 9), ln)
        (y * R(5, 77, 64, seed=2).to(dev)).sum().backward()
        outs.append([t.detach().cpu().clone() for t in (y, a.grad, b.grad, ln.weight.grad, ln.bias.grad)])
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert (outs[0][2] == 0).float().mean() > 0.05          # the mask did drop something


def test_dropout_mask_consistency(dev):
    from tatt_amd import functional as Fh
    Fh.set_seed(dev, 42)
    x = torch.ones(1 << 16, device=dev, requires_grad=True)
    y = Fh.dropout(x, 0.1, True, 7)
    keep = (y != 0).float()
    frac = float(keep.mean())
    assert abs(frac - 0.9) < 0.01, frac
    assert torch.allclose(y[y != 0], torch.full_like(y[y != 0], 1 / 0.9))
    y.sum().backward()
    assert torch.equal((x.grad != 0).float(), keep)           # backward regenerates the same mask
    y2 = Fh.dropout(x, 0.1, True, 8)
    assert not torch.equal((y2 != 0), (y != 0))              # a different site draws a different mask
    Fh.next_dropout_step(dev)
    y3 = Fh.dropout(x, 0.1, True, 7)
    assert not torch.equal((y3 != 0), (y != 0))              # a new step draws a different mask
    assert Fh.dropout(x, 0.1, False, 7) is x


# ------------------------------------------------------------------------------------------- GRUs
def _gru_sd(seed):
    g = torch.nn.GRU(64, 32, bidirectional=True, batch_first=True)
    torch.manual_seed(seed)
    for p in g.parameters():
        torch.nn.init.uniform_(p, -0.3, 0.3)
    return g


@pytest.mark.parametrize("vertical", [True, False])
@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (3, 5, 7)])
def test_bigru32(dev, vertical, B, H, W):
    from tatt_amd import functional as Fh
    g = _gru_sd(3)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
             "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    params = [getattr(g, n).detach() for n in names]
    x = R(B, H, W, 64)

    def ref(x, *ps):
        sd = {"g." + n: p for n, p in zip(names, ps)}
        if vertical:      # sequences along H for every (b, w)
            y = O.bigru(x.permute(0, 2, 1, 3).reshape(B * W, H, 64), sd, "g")
            return y.reshape(B, W, H, 64).permute(0, 2, 1, 3)
        return O.bigru(x.reshape(B * H, W, 64), sd, "g").reshape(B, H, W, 64)

    compare_fn("bigru32", lambda x, *ps: Fh.BiGRU32Fn.apply(x, *ps, vertical), ref, [x] + params, dev, grtol=1e-3)


@pytest.mark.parametrize("vertical,cat", [(True, True), (False, False)])
def test_gru_block_fused(dev, vertical, cat):
    """GruBlock = 1x1 conv (over cat[x, tp_map] for gru1) + BiGRU, as the single fused operator."""
    from tatt_amd import functional as Fh
    from tatt_amd.tsrn import GruBlock
    B, H, W = 2, 16, 64
    torch.manual_seed(11)
    blk = GruBlock(128 if cat else 64, 64)
    names = [n for n, _ in blk.named_parameters()]
    params = [p.detach() for _, p in blk.named_parameters()]
    x, xb = R(B, H, W, 64), (R(B, H, W, 64, seed=1) if cat else None)

    def ref(x, xb, *ps):
        sd = {"b." + n: p for n, p in zip(names, ps)}
        inp = torch.cat([x, xb], -1) if xb is not None else x
        inp = inp.permute(0, 3, 1, 2)                        # NCHW
        if vertical:
            return O.gru_block(inp.transpose(-1, -2), sd, "b").transpose(-1, -2).permute(0, 2, 3, 1)
        return O.gru_block(inp, sd, "b").permute(0, 2, 3, 1)

    def hip(x, xb, *ps):
        b2 = GruBlock(128 if cat else 64, 64).to(dev)
        for (n, _), p in zip(list(b2.named_parameters()), ps):
            mod, leaf = n.rsplit(".", 1)
            setattr(b2.get_submodule(mod), leaf, torch.nn.Parameter(p))
        hip.blk = b2
        return Fh.gru_block(x, b2, vertical, xb=xb)

    compare_fn("gru_block", hip, ref, [x, xb] + params, dev,
               grad_mask=[True, cat] + [False] * len(params), grtol=1e-3)
    # parameter gradients
    cs = [p.clone().requires_grad_(True) for p in params]
    wgt = R(B, H, W, 64, seed=5)
    (ref(x, xb, *cs) * wgt).sum().backward()
    b2 = hip.blk
    for p_ in b2.parameters():
        p_.grad = None
    (Fh.gru_block(x.to(dev), b2, vertical, xb=None if xb is None else xb.to(dev)) * wgt.to(dev)).sum().backward()
    for (n, p), c in zip(b2.named_parameters(), cs):
        check_close("gru_block.grad." + n, p.grad, c.grad, rtol=2e-3, atol=2e-4 * float(c.grad.abs().max()))


@pytest.mark.parametrize("B", [1, 2, 5])
def test_query_gru(dev, B):
    from tatt_amd import functional as Fh
    H, W, C = 8, 8, 64                       # GRU(512 -> 2 x 256)
    g = torch.nn.GRU(C * H, C * H // 2, bidirectional=True, batch_first=True)
    torch.manual_seed(5)
    for p in g.parameters():
        torch.nn.init.uniform_(p, -0.08, 0.08)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
             "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    params = [getattr(g, n).detach() for n in names]
    emb = R(H * W, C)

    def ref(emb, *ps):
        sd = {"ig.transformer.gru_encoding." + n: p for n, p in zip(names, ps)}
        sd["ig.init_factor.weight"] = emb
        return O.query_embedding(sd, "ig", B, H, W).reshape(B, H, W, C)

    compare_fn("query_gru", lambda emb, *ps: Fh.QueryGruFn.apply(emb, *ps, B, H, W), ref, [emb] + params, dev,
               grtol=1e-3)


def _qgru_run(dev, B, fwd_chain, bwd_chain, seed=5, sb=False):
    from tatt_amd import functional as Fh
    H, W, C = 16, 64, 64                     # the published geometry: GRU(1024 -> 2 x 512), 64 rows
    g = torch.nn.GRU(C * H, C * H // 2, bidirectional=True, batch_first=True)
    torch.manual_seed(seed)
    for p in g.parameters():
        torch.nn.init.uniform_(p, -0.05, 0.05)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
             "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    leaves = [R(H * W, C).to(dev).requires_grad_()] + [getattr(g, n).detach().to(dev).requires_grad_() for n in names]
    dq = R(B, H, W, C, seed=3).to(dev)
    old = Fh.QGRU_CHAIN_FWD, Fh.QGRU_CHAIN_BWD, Fh.QGRU_CHAIN_SB, Fh.QGRU_WGRAD_SB
    Fh.QGRU_CHAIN_FWD, Fh.QGRU_CHAIN_BWD, Fh.QGRU_CHAIN_SB, Fh.QGRU_WGRAD_SB = fwd_chain, bwd_chain, sb, sb
    try:
        q = Fh.QueryGruFn.apply(*leaves, B, H, W)
        grads = torch.autograd.grad(q, leaves, dq)
        torch.cuda.synchronize()
        Fh.qgru_chain_check()
    finally:
        Fh.QGRU_CHAIN_FWD, Fh.QGRU_CHAIN_BWD, Fh.QGRU_CHAIN_SB, Fh.QGRU_WGRAD_SB = old
    return (q.detach(),) + tuple(grads)


@pytest.mark.parametrize("B", [2, 5, 48])
def test_query_gru_persistent_chain_matches_stepwise(dev, B):
    """tatt_qgru_fwd_chain / tatt_qgru_bwd_chain (one persistent launch per recurrence; work-groups hand h / dgh to their group through
    write-through stores and flag words) against the per-step launches on the same inputs (model/transformer_v2.py:201-221): the
    backward chain runs the same arithmetic in the same order, the forward chain splits the contraction over 8 waves instead of 4."""
    ref = _qgru_run(dev, B, False, False)
    for fwd_chain, bwd_chain in ((True, False), (False, True), (True, True)):
        out = _qgru_run(dev, B, fwd_chain, bwd_chain)
        for k, (a, b) in enumerate(zip(ref, out)):
            err = float((a - b).abs().max() / (a.abs().max() + 1e-20))
            assert err < 1e-5, (fwd_chain, bwd_chain, k, err)
    # the split-bf16 form of the recurrent products (default): 2^-16 relative per product, carried through B dependent steps
    out = _qgru_run(dev, B, True, True, sb=True)
    worst = 0.0
    for k, (a, b) in enumerate(zip(ref, out)):
        err = float((a - b).abs().max() / (a.abs().max() + 1e-20))
        worst = max(worst, err)
        assert err < 1e-4, ("split-bf16", k, err)
    print("B = %d: split-bf16 chains vs fp32 per-step launches: worst relative error %.2e" % (B, worst))


def test_query_gru_persistent_chain_under_load(dev):
    """The hand-off between work-groups must hold when the chip is busy and the consumers' caches are warm from the previous run
    (MI355X_MICROARCH.md: test every hand-off under uneven load, checking every word): 30 runs of the B = 48 chains beside a second
    stream that keeps streaming kernels in flight, every output word against the first run."""
    ref = _qgru_run(dev, 48, True, True, sb=True)
    side = torch.cuda.Stream()
    x = torch.randn(64 << 20, device=dev)
    for rep in range(30):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 4):
                x.mul_(1.0001)
        out = _qgru_run(dev, 48, True, True, sb=True)
        for k, (a, b) in enumerate(zip(ref, out)):
            assert torch.equal(a, b), (rep, k, float((a - b).abs().max()))
    torch.cuda.synchronize()


@pytest.mark.parametrize("groups,lds", [(32, 0), (64, 0), (64, 65536)])
def test_query_gru_persistent_chain_beside_a_cu_holder(dev, groups, lds):
    """the chains had shared the chip with streaming kernels and a co-running process, never with a kernel that HOLDS
    CUs the way a collective's channel kernels do.  Here 32 / 64 work-groups of 512 threads (with and without 64 KB of LDS each) sit
    resident for ~1 ms on a second stream while the B = 48 forward and backward chains (256 work-groups, one per CU) run: every output
    word must equal the unloaded run, no wait may expire (launch word and the device's sticky word stay 0), and the stall is printed."""
    from tatt_amd import ops, functional as Fh
    ref = _qgru_run(dev, 48, True, True, sb=True)
    sticky = Fh.sticky_word(dev)
    side = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device=dev)

    def timed(load):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if load:
            with torch.cuda.stream(side):
                ops.call("tatt_cu_holder", groups, 100000, lds, ops.P(sink), ops.stream())      # 1 ms
        e0.record()
        out = _qgru_run(dev, 48, True, True, sb=True)
        e1.record()
        torch.cuda.synchronize()
        return out, e0.elapsed_time(e1)
    _, t_alone = timed(False)
    worst = 0.0
    for rep in range(6):
        out, t = timed(True)
        worst = max(worst, t)
        for k, (a, b) in enumerate(zip(ref, out)):
            assert torch.equal(a, b), (rep, k, float((a - b).abs().max()))
    Fh.qgru_chain_check()
    assert int(sticky[0].item()) == 0
    print("query-GRU chains beside %d resident work-groups (%d KB LDS each): %.3f ms alone, worst %.3f ms" % (groups, lds // 1024, t_alone, worst))


# ------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("Lq,S", [(1024, 26), (26, 26), (70, 5)])
def test_mha(dev, Lq, S):
    from tatt_amd import functional as Fh
    B, E = 2, 64
    mha = torch.nn.MultiheadAttention(E, 4, dropout=0.1)
    torch.manual_seed(1)
    torch.nn.init.uniform_(mha.in_proj_bias, -0.2, 0.2)
    torch.nn.init.uniform_(mha.out_proj.bias, -0.2, 0.2)
    ps = [mha.in_proj_weight.detach(), mha.in_proj_bias.detach(), mha.out_proj.weight.detach(),
          mha.out_proj.bias.detach()]
    q, k, v = R(B, Lq, E), R(B, S, E, seed=1), R(B, S, E, seed=2)

    def ref(q, k, v, w, b, ow, ob):
        sd = {"m.in_proj_weight": w, "m.in_proj_bias": b, "m.out_proj.weight": ow, "m.out_proj.bias": ob}
        return O.mha(q, k, v, sd, "m", 4)

    def hip(q, k, v, w, b, ow, ob):
        m = torch.nn.MultiheadAttention(E, 4, dropout=0.1).to(dev)
        m.in_proj_weight, m.in_proj_bias = torch.nn.Parameter(w), torch.nn.Parameter(b)
        m.out_proj.weight, m.out_proj.bias = torch.nn.Parameter(ow), torch.nn.Parameter(ob)
        hip.m = m
        return Fh.multihead_attention(q, k, v, m, False, 0)

    cin = [q, k, v] + ps
    # parameters are re-wrapped inside hip(): compare input grads here, parameter grads below
    compare_fn("mha", hip, ref, cin, dev, grad_mask=[True, True, True, False, False, False, False])
    # parameter gradients
    cs = [t.clone().requires_grad_(True) for t in ps]
    o, w_ = ref(q, k, v, *cs)
    (o.sum() + (w_ * R(*w_.shape, seed=9)).sum()).backward()
    m = hip.m
    for p in m.parameters():
        p.grad = None
    o2, w2 = Fh.multihead_attention(q.to(dev), k.to(dev), v.to(dev), m, False, 0)
    (o2.sum() + (w2 * R(*w_.shape, seed=9).to(dev)).sum()).backward()
    for got, want, n in zip([m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias], cs,
                            ["in_w", "in_b", "out_w", "out_b"]):
        check_close("mha.grad." + n, got.grad, want.grad, rtol=1e-3, atol=1e-4 * float(want.grad.abs().max()))


def test_attn_dropout_fwd_bwd_consistent(dev):
    """With attention dropout on, backward must see the same mask: check d/dV of sum(ctx) == sum of dropped probs."""
    from tatt_amd import functional as Fh
    Fh.set_seed(dev, 7)
    B, Lq, S = 1, 64, 26
    Q, K = R(B, Lq, 64).to(dev), R(B, S, 64, seed=1).to(dev)
    V = R(B, S, 64, seed=2).to(dev).requires_grad_(True)
    ctx, w = Fh.AttnCoreFn.apply(Q, K, V, 0.1, 11)
    ctx.sum().backward()
    # d sum(ctx) / dV[s, h*16+i] = sum_q P_dropped[h, q, s]; head-mean weights w = mean_h P_dropped
    got = V.grad.reshape(S, 4, 16).mean(2).sum(1) / 4          # mean over heads of column sums
    check_close("attn_dropout", got, w[0].sum(0), rtol=1e-4, atol=1e-4)
    zeros = float((w == 0).float().mean())
    assert zeros < 0.01          # head-averaged weights are rarely exactly zero; but mask must have hit some heads
    assert abs(float(w.detach().sum()) / Lq - 1.0) < 0.1


# ------------------------------------------------------------------------------------------- TPS
def test_tps_golden_and_grad(dev):
    from tatt_amd import functional as Fh
    from tatt_amd.tsrn import TPSSpatialTransformer
    z = np.load("tests/golden/tps.npz")
    tps = TPSSpatialTransformer((16, 64), 20, (0.05, 0.05))
    x, ctrl = torch.from_numpy(z["x"]), torch.from_numpy(z["ctrl"])
    sd = {"t." + k: v for k, v in tps.state_dict().items()}

    def ref(x, ctrl):
        y, src = O.tps_transform(x, ctrl, sd, "t")
        return y.permute(0, 2, 3, 1), src

    def hip(x, ctrl):
        t = tps.to(dev)
        src = Fh.TpsGridFn.apply(ctrl, t.inverse_kernel, t.padding_matrix, t.target_coordinate_repr)
        return Fh.GridSampleFn.apply(x, src), src

    # conditioning: the reference evaluates repr @ (inverse_kernel @ Y) in fp32 with entries up to 87 that cancel -- its own
    # result is 4.5e-6 away from the exact coordinates (fp64); the kernel accumulates in fp64.  4.5e-6 * 64 px * (white-noise
    # pixel differences ~1) = up to ~1e-3 in the sampled image, hence the tolerances below.
    compare_fn("tps", hip, ref, [x, ctrl], dev, grad_mask=[False, True], rtol=1e-3, atol=2e-3, grtol=1e-2, gatol=1e-2)
    y, src = hip(x.to(dev), ctrl.to(dev))
    check_close("tps.golden.src", src, torch.from_numpy(z["src"]), rtol=1e-5, atol=5e-5)
    check_close("tps.golden.y", y.permute(0, 3, 1, 2), torch.from_numpy(z["y"]), rtol=1e-3, atol=2e-3)
    # on a smooth image (horizontal ramp) the same coordinates give a tight match -- away from the top/bottom border, where
    # the zero padding makes the output as sensitive to the y coordinate as white noise (d out / d cy = 16 * value)
    xs = torch.linspace(0, 1, 64).reshape(1, 1, 1, 64).expand(3, 4, 16, 64).contiguous()
    ys, srcs = O.tps_transform(xs, ctrl, sd, "t")
    yh, _ = hip(xs.to(dev), ctrl.to(dev))
    inner = ((srcs[..., 1] > 0.06) & (srcs[..., 1] < 0.94) & (srcs[..., 0] > 0.02) & (srcs[..., 0] < 0.98))
    inner = inner.reshape(3, 1, 16, 64).expand(3, 4, 16, 64)
    d = (yh.permute(0, 3, 1, 2).cpu() - ys).abs()
    assert float(d[inner].max()) < 1e-4, float(d[inner].max())
    assert float(d.max()) < 2e-3


# ---- TBSRN variant kernels -------------------------------------------------------------------------
@pytest.mark.parametrize("B,Pn", [(3, 50), (4, 1024), (1, 3)])
def test_tbsrn_layer_norm_mode1(dev, B, Pn):
    """(backward at C = 128 without dropout = ln_bwd_c128_kernel: 16-lane groups, four rows in flight; ragged row counts included)"""
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(11)
    a, b = torch.randn(B, Pn, 128, generator=g), torch.randn(B, Pn, 128, generator=g)
    ga, be = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    compare_fn("tbsrn_ln", lambda a, b, g_, be_: Fh.LayerNormFn.apply(a, b, g_, be_, 1e-6, 1, 0.0, 0),
               lambda a, b, g_, be_: O.tbsrn_layer_norm(a + b, g_, be_), [a, b, ga, be], dev)


def _ref_self_attn(q, k, v, h):
    B, Pn, E = q.shape
    d = E // h
    sp = lambda t: t.reshape(B, Pn, h, d).transpose(1, 2)
    p = torch.softmax(sp(q) @ sp(k).transpose(-2, -1) / d ** 0.5, -1)
    return (p @ sp(v)).transpose(1, 2).reshape(B, Pn, E)


@pytest.mark.parametrize("B,Pn", [(2, 256), (1, 1024), (3, 192)])
def test_self_attention_core(dev, B, Pn):
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(12)
    q, k, v = (torch.randn(B, Pn, 128, generator=g) for _ in range(3))
    compare_fn("self_attn", lambda q, k, v: Fh.SelfAttnCoreFn.apply(q, k, v, 4, 0.0, 0),
               lambda q, k, v: _ref_self_attn(q, k, v, 4), [q, k, v], dev, grtol=1e-3)


def test_self_attention_dropout_is_consistent(dev):
    """Dropout on the probabilities: forward and backward must use the same regenerated mask -- check with the
    linearity of the op in V: out(V) is linear, so <dV, V> == <w, out>."""
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(13)
    q, k = (torch.randn(2, 128, 128, generator=g).to(dev) for _ in range(2))
    v = torch.randn(2, 128, 128, generator=g).to(dev).requires_grad_(True)
    w = torch.randn(2, 128, 128, generator=g).to(dev)
    out = Fh.SelfAttnCoreFn.apply(q, k, v, 4, 0.1, 77)
    (out * w).sum().backward()
    assert abs(float((v.grad * v).sum()) - float((out * w).sum())) < 1e-3 * float((out * w).abs().sum())
    out0 = Fh.SelfAttnCoreFn.apply(q, k, v.detach(), 4, 0.0, 77)
    assert float((out - out0).abs().max()) > 1e-3                  # masks were applied


def test_cat_positional_table(dev):
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(14)
    x, pe = torch.randn(3, 40, 64, generator=g), torch.randn(40, 64, generator=g)
    compare_fn("cat_pe", lambda x, pe: Fh.CatPEFn.apply(x, pe),
               lambda x, pe: torch.cat([x, pe.unsqueeze(0).expand(3, 40, 64)], -1), [x, pe], dev, grad_mask=[True, False])


@pytest.mark.parametrize("B,H,W,Cout,act,beta", [(2, 16, 64, 64, 0, 0.0), (3, 5, 128, 256, 2, 0.0), (1, 1, 64, 64, 0, 0.5),
                                                 (48, 16, 64, 64, 0, 0.0), (7, 16, 64, 128, 1, 0.0)])
def test_conv3_weight_stationary_kernel(dev, B, H, W, Cout, act, beta):
    """tatt_conv3_c64_fwd_ws16 (exact-fp32 weight-stationary kernel) against F.conv2d, forward filter (repack mode 6)."""
    from tatt_amd import ops
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, H, W, 64, generator=g)
    w = torch.randn(Cout, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    y0 = torch.randn(B, H, W, Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    ref = {0: lambda t: t, 1: torch.relu, 2: lambda t: O.mish(t.float()).double()}[act](ref) + beta * y0.double()
    xd, wd, bd, yd = x.to(dev), w.to(dev), b.to(dev), y0.to(dev).clone()
    for entry, mode in (("tatt_conv3_c64_fwd_ws16", 6),):
        yd = y0.to(dev).clone()
        wl = ops.repack_weight(wd, mode)
        ops.call(entry, ops.P(xd), ops.P(wl), ops.P(bd), ops.P(yd), B, H, W, Cout, act, beta, ops.stream())
        check_close(entry, yd, ref.float(), 2e-4, 2e-4)


def test_conv3_weight_stationary_dgrad(dev):
    """repack mode 7: the data gradient of a 64-output-channel 3x3 convolution (Cin = 64 and 256), exact fp32."""
    from tatt_amd import ops
    g = torch.Generator().manual_seed(22)
    for Cin in (64, 256):
        x = torch.randn(2, 16, 64, Cin, generator=g).permute(0, 3, 1, 2).requires_grad_(True)
        w = torch.randn(64, Cin, 3, 3, generator=g) * 0.05
        dy = torch.randn(2, 16, 64, 64, generator=g)
        torch.nn.functional.conv2d(x, w, None, padding=1).backward(dy.permute(0, 3, 1, 2))
        dx = ops.conv2d_dgrad(dy.to(dev), w.to(dev))
        check_close("conv3_ws_dgrad_%d" % Cin, dx, x.grad.permute(0, 2, 3, 1), 2e-4, 2e-4)
        dyd, wd = dy.to(dev), w.to(dev)                # keep the operands alive while the kernels run
        for entry, mode in (("tatt_conv3_c64_fwd_ws16", 7),):
            dxe = torch.empty(2, 16, 64, Cin, device=dev)
            wl = ops.repack_weight(wd, mode)
            ops.call(entry, ops.P(dyd), ops.P(wl), None, ops.P(dxe), 2, 16, 64, Cin, 0, 0.0, ops.stream())
            check_close("%s_dgrad_%d" % (entry, Cin), dxe, x.grad.permute(0, 2, 3, 1), 2e-4, 2e-4)


@pytest.mark.parametrize("B,C,H,W,nhwc", [(3, 4, 32, 128, True), (2, 3, 8, 16, False), (5, 4, 64, 256, True)])
def test_image_loss_fused(dev, B, C, H, W, nhwc):
    """tatt_image_loss_fwd/bwd against the oracle's ImageLoss (reference loss/image_loss.py) and its autograd gradient: per-sample
    form and the scaled batch mean; SR over NHWC memory (what the generator returns) and plain NCHW."""
    from tatt_amd.train import image_loss, image_loss_mean
    g = torch.Generator().manual_seed(41)
    sr = (torch.rand(B, C, H, W, generator=g) * 2 - 1).requires_grad_(True)
    hr = torch.rand(B, C, H, W, generator=g)
    wts = torch.rand(B, generator=g)
    ref = O.image_loss(sr, hr)
    (ref * wts).sum().backward()
    srd = sr.detach().to(dev)
    if nhwc:
        srd = srd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    srd.requires_grad_(True)
    got = image_loss(srd, hr.to(dev))
    check_close("image_loss", got, ref, 1e-5, 1e-7)
    (got * wts.to(dev)).sum().backward()
    check_close("image_loss_grad", srd.grad, sr.grad, 2e-4, 1e-9)
    srd.grad = None
    sr.grad = None
    (O.image_loss(sr, hr).mean() * 100).backward()
    m = image_loss_mean(srd, hr.to(dev), scale=100.0)
    assert abs(float(m) - float(ref.mean() * 100)) < 1e-5 * float(ref.mean() * 100)
    m.backward()
    check_close("image_loss_mean_grad", srd.grad, sr.grad, 2e-4, 1e-9)


def test_image_loss_requires_gpu():
    from tatt_amd.train import image_loss
    with pytest.raises(RuntimeError, match="GPU"):
        image_loss(torch.rand(1, 4, 8, 8), torch.rand(1, 4, 8, 8))


def test_semantic_loss_and_psnr_kernels(dev):
    """tatt_semantic_loss_fwd/bwd, tatt_psnr against the reference-generated vectors (tests/golden/losses.npz)."""
    from tatt_amd.train import semantic_loss, calculate_psnr
    z = np.load("tests/golden/losses.npz")
    pred = torch.from_numpy(z["pred"]).to(dev).requires_grad_(True)
    loss = semantic_loss(pred, torch.from_numpy(z["gt"]).to(dev))
    assert abs(float(loss.detach()) - float(z["sem"])) < 1e-6
    (loss * 3.0).backward()
    check_close("semantic_loss_grad", pred.grad, 3.0 * torch.from_numpy(z["dpred"]), 1e-5, 1e-9)
    a, b = torch.from_numpy(z["a"]).to(dev), torch.from_numpy(z["b"]).to(dev)
    assert abs(float(calculate_psnr(a, b)) - float(z["psnr"])) < 1e-4
    a_nhwc = a.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)                 # the generator's output layout
    assert abs(float(calculate_psnr(a_nhwc, b)) - float(z["psnr"])) < 1e-4
    assert float(calculate_psnr(a, a)) == float("inf")


@pytest.mark.parametrize("sb", [True, False])
@pytest.mark.parametrize("B,H,W", [(2, 32, 128), (3, 16, 64), (1, 8, 64), (1, 4, 64), (5, 64, 256)])
def test_conv9_mfma_toeplitz(dev, B, H, W, sb, monkeypatch):
    """tatt_conv9_c64_to_c4_sb / _mfma (4 pixels x 4 channels per MFMA column block, Toeplitz-expanded filter; split-bf16 products on
    the bf16 matrix cores -- the default -- and exact fp32 ones): the 64->4 reconstruction convolution (repack mode 12 / 8) and the
    data gradient of a 4->64 convolution (mode 13 / 9) against F.conv2d in fp64."""
    from tatt_amd import ops
    monkeypatch.setattr(ops, "CONV9_SB", sb)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, H, W, 64, generator=g)
    w = torch.randn(4, 64, 9, 9, generator=g) * 0.02
    b = torch.randn(4, generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=4).permute(0, 2, 3, 1)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y = ops.conv2d_forward(xd, wd, bd)
    check_close("conv9_mfma_fwd", y, ref.float(), 2e-4, 2e-4)
    # against the scale of the result: the split-bf16 form keeps 16 mantissa bits per operand (2^-16 relative per product)
    err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < (2e-5 if sb else 3e-6), err
    assert torch.equal(y, ops.conv2d_forward(xd, wd, bd))
    # data gradient of block1 (4 -> 64): dx = conv_transpose(dy, w1)
    w1 = (torch.randn(64, 4, 9, 9, generator=g) * 0.02)
    xin = torch.randn(B, 4, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    dy = torch.randn(B, H, W, 64, generator=g)
    torch.nn.functional.conv2d(xin, w1.double(), None, padding=4).backward(dy.permute(0, 3, 1, 2).double())
    dyd, w1d = dy.to(dev), w1.to(dev)
    dx = ops.conv2d_dgrad(dyd, w1d)
    check_close("conv9_mfma_dgrad", dx, xin.grad.permute(0, 2, 3, 1).float(), 2e-4, 2e-4)
    gref = xin.grad.permute(0, 2, 3, 1)
    err = float((dx.cpu().double() - gref).abs().max() / gref.abs().max())
    assert err < (2e-5 if sb else 3e-6), err
    # the packed-filter cache follows an in-place update of the weights
    wd.mul_(2.0)
    check_close("conv9_mfma_fwd_after_update", ops.conv2d_forward(xd, wd, bd), (2 * (ref - b.double()) + b.double()).float(), 4e-4, 4e-4)


@pytest.mark.parametrize("sb", [True, False])
@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (3, 4, 64), (1, 8, 192), (67, 8, 64), (2, 32, 128)])
def test_conv9_c4_to_c64_weight_stationary(dev, B, H, W, sb, monkeypatch):
    """tatt_conv9_c4_to_c64_sb (k = 8 taps x 4 input channels on the bf16 matrix cores, split operands; the default) and
    tatt_conv9_c4_to_c64 (k = the 4 input channels, 81 filter registers per lane, exact fp32): block1's 4->64 convolution with bias
    (reference model/tsrn.py:597) and the data gradient of the 64->4 reconstruction convolution (:623) against F.conv2d in fp64.
    (67, 8, 64): more tiles than persistent groups' first round can take evenly."""
    from tatt_amd import ops
    monkeypatch.setattr(ops, "CONV9_SB", sb)
    g = torch.Generator().manual_seed(53)
    x = torch.randn(B, H, W, 4, generator=g)
    w = torch.randn(64, 4, 9, 9, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=4).permute(0, 2, 3, 1)
    y = ops.conv2d_forward(x.to(dev), w.to(dev), b.to(dev))
    check_close("conv9_4_64_fwd", y, ref.float(), 1e-4, 1e-4)
    err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < (2e-5 if sb else 3e-6), err
    y = ops.conv2d_forward(x.to(dev), w.to(dev), b.to(dev), act=ops.ACT_RELU)
    check_close("conv9_4_64_fwd_relu", y, ref.clamp_min(0).float(), 1e-4, 1e-4)
    w2 = torch.randn(4, 64, 9, 9, generator=g) * 0.02
    xin = torch.randn(B, 64, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    dy = torch.randn(B, H, W, 4, generator=g)
    torch.nn.functional.conv2d(xin, w2.double(), None, padding=4).backward(dy.permute(0, 3, 1, 2).double())
    dx = ops.conv2d_dgrad(dy.to(dev), w2.to(dev))
    check_close("conv9_64_4_dgrad", dx, xin.grad.permute(0, 2, 3, 1).float(), 1e-4, 1e-4)
    gref = xin.grad.permute(0, 2, 3, 1)
    err = float((dx.cpu().double() - gref).abs().max() / gref.abs().max())
    assert err < (2e-5 if sb else 3e-6), err


@pytest.mark.parametrize("sb", [True, False])
@pytest.mark.parametrize("B,H,W", [(2, 32, 128), (3, 4, 64), (1, 8, 192), (67, 8, 64), (2, 64, 256)])
def test_conv9_wgrad_mfma(dev, B, H, W, sb, monkeypatch):
    """tatt_conv9_c64_c4_wgrad_sb (split-bf16 products, the default) / tatt_conv9_c64_c4_wgrad (exact fp32): weight gradient of the 64->4 reconstruction convolution (reference model/tsrn.py:623) with the
    Toeplitz expansion on the dy operand, against autograd in fp64.  Shapes cover one tile, a ragged tile count over the persistent
    groups (67 x 2 tiles), three pixel chunks per row and the large-tile geometry."""
    from tatt_amd import ops
    monkeypatch.setattr(ops, "CONV9_SB", sb)
    tol = 2e-5 if sb else 2e-6               # 2^-16 relative per product against exact fp32 products
    g = torch.Generator().manual_seed(47)
    x = torch.randn(B, H, W, 64, generator=g)
    dy = torch.randn(B, H, W, 4, generator=g)
    w = torch.zeros(4, 64, 9, 9, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w, None, padding=4).backward(dy.permute(0, 3, 1, 2).double())
    dw = ops.conv_wgrad(x.to(dev), dy.to(dev), 4, 9, 9)
    scale = float(w.grad.abs().max())
    err = float((dw.double().cpu() - w.grad).abs().max())
    assert err <= tol * scale * max(1.0, (B * H * W / 8192) ** 0.5), (err, scale)
    assert torch.equal(dw, ops.conv_wgrad(x.to(dev), dy.to(dev), 4, 9, 9))          # deterministic
    # the same kernel with the tensors' roles exchanged: weight gradient of block1's 4 -> 64 convolution (model/tsrn.py:597)
    x4 = torch.randn(B, H, W, 4, generator=g)
    dy64 = torch.randn(B, H, W, 64, generator=g)
    w1 = torch.zeros(64, 4, 9, 9, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x4.permute(0, 3, 1, 2).double(), w1, None, padding=4).backward(dy64.permute(0, 3, 1, 2).double())
    dw1 = ops.conv_wgrad(x4.to(dev), dy64.to(dev), 64, 9, 9)
    scale = float(w1.grad.abs().max())
    err = float((dw1.double().cpu() - w1.grad).abs().max())
    assert err <= tol * scale * max(1.0, (B * H * W / 8192) ** 0.5), (err, scale)


# ------------------------------------------------------------------------------------------- conv + BatchNorm folding
@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (5, 3, 128)])
def test_conv_bn_folded_equals_operator_chain(dev, B, H, W):
    """conv -> bn -> mish -> conv -> bn with the statistics taken from the convolution's epilogue and bn + mish applied while the
    next convolution stages its input (ConvBnFn / BatchNormApplyFn) == the operator chain conv2d | batch_norm_act | conv2d |
    batch_norm_act (reference model/tsrn.py:877-886): outputs, running statistics, every gradient."""
    from tatt_amd import functional as Fh
    from tatt_amd.tsrn import RecurrentResidualBlock
    from tatt_amd.ops import ACT_MISH, ACT_NONE
    torch.manual_seed(5)
    res = []
    for folded in (True, False):
        torch.manual_seed(5)
        blk = RecurrentResidualBlock(64, 0).to(dev).train()
        with torch.no_grad():
            for bn in (blk.bn1, blk.bn2):
                bn.weight.add_(0.3 * R(64).to(dev))
                bn.bias.add_(0.3 * R(64, seed=1).to(dev))
        x = (R(B, H, W, 64, seed=2) + 0.5).to(dev).requires_grad_(True)
        if folded:
            y1, st1 = Fh.conv_bn(x, blk.conv1, blk.bn1)
            y2, st2 = Fh.conv_bn(y1, blk.conv2, blk.bn2, prev=(st1, blk.bn1, ACT_MISH))
            out = Fh.bn_apply_stats(y2, st2, blk.bn2, ACT_NONE)
        else:
            r = Fh.conv2d(x, blk.conv1.weight, blk.conv1.bias)
            r = Fh.batch_norm_act(r, blk.bn1, ACT_MISH, False)
            r = Fh.conv2d(r, blk.conv2.weight, blk.conv2.bias)
            out = Fh.batch_norm_act(r, blk.bn2, ACT_NONE, False)
        (out * R(B, H, W, 64, seed=3).to(dev)).sum().backward()
        d = {"out": out, "dx": x.grad}
        for n in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "bn1.weight", "bn1.bias", "bn2.weight", "bn2.bias"):
            d["g." + n] = blk.get_parameter(n).grad
        for n in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var"):
            d[n] = blk.get_buffer(n)
        res.append({k: v.detach().float().cpu().clone() for k, v in d.items()})
    bad = []
    for k in res[0]:
        ref = float(res[1][k].abs().max()) + 1e-12
        noise = k in ("g.conv1.bias", "g.conv2.bias")          # mathematically zero (a bias in front of a BatchNorm): round-off only
        err = float((res[0][k] - res[1][k]).abs().max()) / (1.0 if noise else ref)
        if not err < (1e-3 if noise else 2e-4):
            bad.append("%s: %.3e" % (k, err))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (5, 3, 128), (48, 16, 64)])
def test_srb_trunk_fused_backward_equals_operator_chain(dev, B, H, W):
    """SrbTrunkFn -- conv -> bn -> mish -> conv -> bn (reference model/tsrn.py:877-886) as one operator whose backward folds both
    BatchNorm backwards into the data-gradient convolutions (partials from the epilogue, the affine form dx = a du + b x + c applied
    while the next convolution stages its input; tatt_conv3_c64_dgrad_bn_sb, tatt_bn_bwd_finish, tatt_bn_bwd_affine) == the operator
    chain conv2d | batch_norm_act | conv2d | batch_norm_act: output, running statistics, every gradient; with and without the deferred
    parameter-gradient lane (the Trainer's mode)."""
    from tatt_amd import functional as Fh
    from tatt_amd.tsrn import RecurrentResidualBlock
    from tatt_amd.ops import ACT_MISH, ACT_NONE
    res = []
    for mode in ("fused", "fused_deferred", "chain"):
        torch.manual_seed(5)
        blk = RecurrentResidualBlock(64, 0).to(dev).train()
        with torch.no_grad():
            for bn in (blk.bn1, blk.bn2):
                bn.weight.add_(0.3 * R(64).to(dev))
                bn.bias.add_(0.3 * R(64, seed=1).to(dev))
        x = (R(B, H, W, 64, seed=2) + 0.5).to(dev).requires_grad_(True)
        if mode != "chain":
            out = Fh.srb_trunk(x, blk)
        else:
            r = Fh.conv2d(x, blk.conv1.weight, blk.conv1.bias)
            r = Fh.batch_norm_act(r, blk.bn1, ACT_MISH, False)
            r = Fh.conv2d(r, blk.conv2.weight, blk.conv2.bias)
            out = Fh.batch_norm_act(r, blk.bn2, ACT_NONE, False)
        loss = (out * R(B, H, W, 64, seed=3).to(dev)).sum()
        if mode == "fused_deferred":
            Fh.SIDE.enabled, Fh.SIDE.stage = True, 0
            try:
                loss.backward()
                assert blk.conv1.weight.grad is None               # registered, not computed, while the main lane runs
                Fh.SIDE.flush()
            finally:
                Fh.SIDE.enabled = False
                Fh.SIDE.release()
        else:
            loss.backward()
        d = {"out": out, "dx": x.grad}
        for n in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "bn1.weight", "bn1.bias", "bn2.weight", "bn2.bias"):
            d["g." + n] = blk.get_parameter(n).grad
        for n in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var"):
            d[n] = blk.get_buffer(n)
        res.append({k: v.detach().float().cpu().clone() for k, v in d.items()})
    bad = []
    for k in res[0]:
        ref = float(res[2][k].abs().max()) + 1e-12
        noise = k in ("g.conv1.bias", "g.conv2.bias")          # mathematically zero (a bias in front of a BatchNorm): round-off only
        for name, got in (("fused", res[0]), ("deferred", res[1])):
            err = float((got[k] - res[2][k]).abs().max()) / (1.0 if noise else ref)
            if not err < (1e-6 * B * H * W + 1e-4 if noise else 2e-4):    # (noise: a sum of B H W round-offs of unit-scale terms)
                bad.append("%s %s: %.3e" % (name, k, err))
    assert not bad, "\n".join(bad)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k                # the lanes change the schedule, not the arithmetic


def test_bn_backward_pieces_vs_fp64(dev):
    """tatt_bn_bwd_partials | tatt_bn_bwd_finish | tatt_bn_bwd_affine (the train-mode BatchNorm backward in pieces, with Mish) against
    autograd in fp64: dgamma, dbeta and dx = a du + b x + c."""
    from tatt_amd import ops
    from tatt_amd.ops import ACT_MISH
    M, C = 3000, 64
    x, dy = R(M, C, seed=1) * 1.5 + 0.3, R(M, C, seed=2)
    gamma, beta = R(C, seed=3) * 0.3 + 1.0, R(C, seed=4) * 0.3
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    xh = (xd - mean) / torch.sqrt(var + 1e-5)
    u = gd * xh + bd
    (u * torch.tanh(torch.nn.functional.softplus(u)) * dy.double()).sum().backward()
    d = lambda t: t.to(dev)
    mean_f, rstd_f = d(mean.detach().float()), d((1.0 / torch.sqrt(var + 1e-5)).detach().float())
    part, G = ops.bn_bwd_partials(d(x), d(dy), mean_f, rstd_f, d(gamma), d(beta), ACT_MISH)
    dg, db, coef = ops.bn_bwd_finish(part, G, C, M, mean_f, rstd_f, d(gamma))
    # du = dy * mish'(u) in fp32 on the host (the affine kernel takes du)
    uu = (gamma * ((x - mean.detach().float()) * (1.0 / torch.sqrt(var + 1e-5)).detach().float()) + beta).double().requires_grad_(True)
    (uu * torch.tanh(torch.nn.functional.softplus(uu))).sum().backward()
    du = (dy.double() * uu.grad).float()
    dx = ops.bn_bwd_affine(d(x), d(du), coef)
    for name, got, want in (("dgamma", dg, gd.grad), ("dbeta", db, bd.grad), ("dx", dx, xd.grad)):
        err = float((got.cpu().double() - want).abs().max() / want.abs().max())
        assert err < 2e-5, (name, err)


# ------------------------------------------------------------------------------------------- split-bf16 3x3 convolution
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 64, 64, 64), (3, 5, 128, 64, 256), (2, 16, 64, 256, 64), (1, 1, 64, 128, 128)])
def test_conv3_split_bf16_vs_fp64(dev, B, H, W, Cin, Cout):
    """tatt_conv3_c64_fwd_sb (bf16 matrix cores, hi/lo operand split, three products, fp32 accumulation) against the fp64 convolution:
    forward and data gradient, also for contractions wider than 64 channels (chunked).  Error model: operands keep 16 mantissa bits,
    so each product is 2^-16 = 1.5e-5 relative at worst and the sum of K = 9 Cin random-sign terms lands around 1e-6 of the output
    scale -- the exact-fp32 kernel sits at 1e-7; both far inside the parity budget (tools/split_bf16_probe.py)."""
    from tatt_amd import ops
    x = R(B, H, W, Cin, seed=1)
    w = R(Cout, Cin, 3, 3, seed=2) * (1.0 / math.sqrt(9 * Cin))
    b = R(Cout, seed=3)
    dy = R(B, H, W, Cout, seed=4)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    ref_dx = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    res = {}
    for sb in (True, False):
        ops.CONV3_SB = sb
        try:
            y = ops.conv2d_forward(x.to(dev), w.to(dev), b.to(dev))
            dx = ops.conv2d_dgrad(dy.to(dev), w.to(dev))
        finally:
            ops.CONV3_SB = True
        res[sb] = (float((y.cpu().double() - ref).abs().max() / ref.abs().max()), float((dx.cpu().double() - ref_dx).abs().max() / ref_dx.abs().max()))
    print("conv3 %dx%dx%d %d->%d: split-bf16 rel-max err fwd %.2e dgrad %.2e | fp32 MFMA fwd %.2e dgrad %.2e" % (
        B, H, W, Cin, Cout, res[True][0], res[True][1], res[False][0], res[False][1]))
    assert res[True][0] < 1e-5 and res[True][1] < 1e-5, res
    assert res[False][0] < 2e-6 and res[False][1] < 2e-6, res


def test_conv3_split_bf16_bn_folding(dev):
    """The split-bf16 kernel with the producer's BatchNorm + mish folded into its input staging and the batch statistics of its own
    output taken from the epilogue == the same through the exact-fp32 kernel (to the split's 1e-5)."""
    from tatt_amd import ops
    B, H, W = 3, 16, 64
    x, w, b = R(B, H, W, 64, seed=1).to(dev), (R(64, 64, 3, 3, seed=2) / 24).to(dev), R(64, seed=3).to(dev)
    sc, sh = (1.0 + 0.3 * R(64, seed=4)).to(dev), (0.2 * R(64, seed=5)).to(dev)
    out = {}
    for sb in (True, False):
        ops.CONV3_SB = sb
        try:
            y, part, G = ops.conv3_bn_forward(x, w, b, sc, sh, ops.ACT_MISH, True)
        finally:
            ops.CONV3_SB = True
        out[sb] = (y.cpu(), part.reshape(G, 2, 64).sum(0).cpu())
    ref = F.conv2d(F.mish(x.cpu() * sc.cpu() + sh.cpu()).permute(0, 3, 1, 2), w.cpu(), b.cpu(), padding=1).permute(0, 2, 3, 1)
    for sb in (True, False):
        y, st = out[sb]
        assert float((y - ref).abs().max() / ref.abs().max()) < (2e-5 if sb else 3e-6), sb
        check_close("stats.sum", st[0].float(), y.double().reshape(-1, 64).sum(0).float(), rtol=1e-5, atol=1e-3)
        check_close("stats.sq", st[1].float(), (y.double() ** 2).reshape(-1, 64).sum(0).float(), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("B,H,W", [(3, 16, 64), (2, 32, 128), (1, 4, 16), (5, 8, 48), (2, 4, 64)])
def test_conv3_split_bf16_square_tiles_vs_fp64_and_row_tiles(dev, B, H, W):
    """Round 6's 4 x 16-pixel-tile kernels (tatt_conv3_sb_generation 4: 32-channel MFMA waves on v_mfma_f32_32x32x16_bf16 with staging
    waves beside them, filter packing 14 / 15; 3: the same without staging waves; both: contraction split over wave pairs, LDS
    exchange, swapped MFMA operands, buffer-descriptor padding, epilogue re-threaded over pixels) against fp64 AND against the
    row-tile kernel of rounds 3-5 on every variant the model runs:
    plain / bias + activation epilogue, Cout = 256, a 64-channel slice of a 256-channel input with beta = 1, BatchNorm + mish folded
    into the staging with output statistics, and the data gradient with the BatchNorm backward folded in on both sides.  Geometries:
    the benchmark's, the large tile, a single tile (every halo side is padding), a ragged tile count, one tile row."""
    from tatt_amd import ops
    from tatt_amd._lib import LIB
    x, w, b = R(B, H, W, 64, seed=1), R(64, 64, 3, 3, seed=2) / 24, R(64, seed=3)
    w256, x256 = R(256, 64, 3, 3, seed=4) / 24, R(B, H, W, 256, seed=5)
    sc, sh = 1.0 + 0.3 * R(64, seed=6), 0.2 * R(64, seed=7)
    coef = torch.stack([1.0 + 0.2 * R(64, seed=8), 0.1 * R(64, seed=9), 0.1 * R(64, seed=10)])
    mean, rstd = 0.1 * R(64, seed=11), 1.0 + 0.2 * R(64, seed=12).abs()
    x2, below = R(B, H, W, 64, seed=13), R(B, H, W, 64, seed=14)
    d = lambda t: t.to(dev)                                               # noqa: E731
    conv = lambda xx, ww, bb=None: F.conv2d(xx.double().permute(0, 3, 1, 2), ww.double(), None if bb is None else bb.double(), padding=1).permute(0, 2, 3, 1)  # noqa: E731
    dconv = lambda g, ww: torch.nn.grad.conv2d_input((B, ww.shape[1], H, W), ww.double(), g.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)  # noqa: E731
    xh = (below.double() - mean.double()) * rstd.double()
    gin = coef[0].double() * x.double() + coef[1].double() * x2.double() + coef[2].double()
    ref = {
        "plain": conv(x, w, b),
        "mish_epilogue": O.mish(conv(x, w, b).float()).double(),
        "cout256": conv(x, w256),
        "dgrad256": dconv(x256, w256),                                   # four 64-channel slices, beta = 1 after the first
        "bn_mish": conv(F.mish(x.double() * sc.double() + sh.double()), w, b),
        "dgrad_in2_epbn": dconv(gin, w) * _mish_grad64(sc.double() * xh + sh.double()),
        "dgrad_in2": dconv(gin, w),
    }
    got = {}
    gens = (4, 3, 1) if W % 64 == 0 else (4, 3)                         # (the row-tile kernel walks 64-pixel segments)
    ops.CONV3_SB_NARROW_MAPS = True                                     # (the W = 16 / 48 cases; off in the product: ops.py)
    for gen in gens:
        ops.CONV3_SB_GENERATION = gen
        try:
            o = {}
            o["plain"] = ops.conv2d_forward(d(x), d(w), d(b))
            o["mish_epilogue"] = ops.conv2d_forward(d(x), d(w), d(b), act=ops.ACT_MISH)
            o["cout256"] = ops.conv2d_forward(d(x), d(w256), None)
            o["dgrad256"] = ops.conv2d_dgrad(d(x256), d(w256))
            y, part, G = ops.conv3_bn_forward(d(x), d(w), d(b), d(sc), d(sh), ops.ACT_MISH, True)
            o["bn_mish"], o["bn_mish.stats"] = y, part.reshape(G, 2, 64).sum(0)
            y, part, G = ops.conv3_dgrad_bn(d(x), d(w), d(x2), d(coef), (d(below), d(mean), d(rstd), d(sc), d(sh), ops.ACT_MISH))
            o["dgrad_in2_epbn"], o["dgrad_in2_epbn.stats"] = y, part.reshape(G, 2, 64).sum(0)
            o["dgrad_in2"] = ops.conv3_dgrad_bn(d(x), d(w), d(x2), d(coef), None)[0]
            got[gen] = {k: v.cpu().double() for k, v in o.items()}
        finally:
            ops.CONV3_SB_GENERATION = 4
            if gen == gens[-1]:
                ops.CONV3_SB_NARROW_MAPS = False
    for k, r in ref.items():
        for gen in gens:
            err = float((got[gen][k] - r).abs().max() / r.abs().max())
            assert err < 2e-5, (k, gen, err)
        # the kernels evaluate the same products and differ in summation order only
        for gen in gens[1:]:
            assert float((got[gens[0]][k] - got[gen][k]).abs().max() / r.abs().max()) < 2e-6, (k, gen)
    for gen in gens[:2]:
        y = got[gen]["bn_mish"]
        check_close("stats.sum", got[gen]["bn_mish.stats"][0].float(), y.reshape(-1, 64).sum(0).float(), rtol=1e-5, atol=1e-3)
        check_close("stats.sq", got[gen]["bn_mish.stats"][1].float(), (y ** 2).reshape(-1, 64).sum(0).float(), rtol=1e-5, atol=1e-3)
        y = got[gen]["dgrad_in2_epbn"]
        check_close("bwd stats.sum", got[gen]["dgrad_in2_epbn.stats"][0].float(), y.reshape(-1, 64).sum(0).float(), rtol=1e-5, atol=1e-3)
        check_close("bwd stats.xhat", got[gen]["dgrad_in2_epbn.stats"][1].float(), (y * xh).reshape(-1, 64).sum(0).float(), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("B,H,W", [(2, 16, 50), (3, 8, 25), (2, 4, 26), (1, 4, 7)])
def test_conv3_split_bf16_ragged_width_relu_and_chunked_contractions(dev, B, H, W):
    """The generation-4 kernel on maps whose width is not a multiple of 16 (the CRNN's 50-, 25-, 26-pixel maps, model/crnn/crnn.py:29-92:
    the last tile column is cut by the map -- requests beyond it are out of range, its pixels are neither stored nor counted), with the
    ReLU output activation (on the LAST chunk of a chunked contraction only) and contractions / outputs of several 64-channel blocks,
    and the ragged weight gradient (tatt_conv3_c64_wgrad_partial_sb, generation 2), against fp64."""
    from tatt_amd import ops
    g = torch.Generator().manual_seed(77 + W)
    for Cin, Cout in ((64, 128), (128, 256), (256, 64)):
        x = torch.randn(B, H, W, Cin, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        b = torch.randn(Cout, generator=g)
        dy = torch.randn(B, H, W, Cout, generator=g)
        wd = w.double().requires_grad_(True)
        bd = b.double().requires_grad_(True)
        xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
        ref = F.conv2d(xd, wd, bd, padding=1)
        ref.backward(dy.double().permute(0, 3, 1, 2))
        refy = ref.detach().permute(0, 2, 3, 1)
        for act, want in ((ops.ACT_NONE, refy), (ops.ACT_RELU, torch.relu(refy))):
            y = ops.conv2d_forward(x.to(dev), w.to(dev), b.to(dev), act, any_width=True)
            err = float((y.cpu().double() - want).abs().max() / refy.abs().max())
            assert err < 2e-5, (Cin, Cout, act, err)
        dx = ops.conv2d_dgrad(dy.to(dev), w.to(dev), any_width=True)
        err = float((dx.cpu().double() - xd.grad.permute(0, 2, 3, 1)).abs().max() / xd.grad.abs().max())
        assert err < 2e-5, (Cin, Cout, "dgrad", err)
        dw, db = ops.conv_wgrad(x.to(dev), dy.to(dev), Cout, 3, 3, want_db=True, any_width=True)
        err = float((dw.cpu().double() - wd.grad).abs().max() / wd.grad.abs().max())
        assert err < 2e-5, (Cin, Cout, "wgrad", err)
        errb = float((db.cpu().double() - bd.grad).abs().max() / bd.grad.abs().max())
        assert errb < 2e-5, (Cin, Cout, "bias gradient", errb)


def _mish_grad64(u):
    u = u.clone().requires_grad_(True)
    F.mish(u).sum().backward()
    return u.grad


# ------------------------------------------------------------------------------------------- score-free self-attention (TBSRN)
@pytest.fixture
def sattn_generation(request):
    """Select the self-attention kernels for one test: 2 = split bf16 (csrc/sattn2.hip, the default), 1 = exact fp32 (csrc/sattn.hip)."""
    from tatt_amd import functional as Fh
    old = Fh.SATTN_SB
    Fh.SATTN_SB = int(request.param) == 2
    yield int(request.param)
    Fh.SATTN_SB = old


@pytest.mark.parametrize("sattn_generation", [2, 1], indirect=True)
@pytest.mark.parametrize("B,Pn", [(2, 256), (1, 1024), (3, 64), (2, 4096)])
def test_flash_self_attention_vs_reference(dev, B, Pn, sattn_generation):
    """csrc/sattn2.hip / csrc/sattn.hip (online-softmax forward, recomputing backward; no (B,h,P,P) tensor) against float64 torch
    attention (reference model/tbsrn.py:130-151), dropout off: values and the three input gradients, relative to the largest entry.
    Generation 1 is exact fp32 (bound 5e-6); generation 2 computes every product as three bf16 products of hi / lo halves (2^-16 per
    product: bound 1e-4, measured 1.5e-5 on the values and 5e-5 on dK / dV); P = 64 runs the fp32 kernels under either setting."""
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(12 + Pn)
    q, k, v, w = (torch.randn(B, Pn, 128, generator=g) for _ in range(4))
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = _ref_self_attn(qd, kd, vd, 4)
    (ref * w.double()).sum().backward()
    qg, kg, vg = (t.clone().to(dev).requires_grad_(True) for t in (q, k, v))
    out = Fh.SelfAttnFlashFn.apply(qg, kg, vg, 4, 0.0, 0)
    (out * w.to(dev)).sum().backward()
    bound = 1e-4 if (sattn_generation == 2 and Pn % 128 == 0) else 5e-6
    for name, got, want in (("out", out, ref), ("dq", qg.grad, qd.grad), ("dk", kg.grad, kd.grad), ("dv", vg.grad, vd.grad)):
        err = float((got.detach().cpu().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert err < bound, (name, err)


@pytest.mark.parametrize("sattn_generation", [2, 1], indirect=True)
@pytest.mark.parametrize("B,Pn", [(2, 128), (1, 1024)])
def test_flash_self_attention_equals_materialised_path_with_dropout(dev, B, Pn, sattn_generation):
    """Dropout ON: the score-free kernels draw the masks of the materialised path (tatt_softmax_rows_fwd: same seed word, site and
    flat index), so outputs and gradients agree to the arithmetic's round-off -- forward and both backward kernels included; with and
    without the keep bits handed from the forward to the backward (SATTN_KEEP_BITS: the same masks, so dK, dV and the values are
    bit-identical and dQ differs by the order of one multiplication)."""
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(3)
    base = [torch.randn(B, Pn, 128, generator=g) for _ in range(3)]
    w = torch.randn(B, Pn, 128, generator=g).to(dev)
    res = []
    for fn, keep_bits in ((Fh.SelfAttnFlashFn, True), (Fh.SelfAttnFlashFn, False), (Fh.SelfAttnCoreFn, True)):
        Fh.set_seed(dev, 5)
        Fh.begin_training_forward(dev)
        Fh.SATTN_KEEP_BITS = keep_bits
        try:
            q, k, v = (t.clone().to(dev).requires_grad_(True) for t in base)
            out = fn.apply(q, k, v, 4, 0.1, 77)
            (out * w).sum().backward()
        finally:
            Fh.SATTN_KEEP_BITS = True
        res.append([t.detach().cpu() for t in (out, q.grad, k.grad, v.grad)])
    bound = 1e-4 if sattn_generation == 2 else 2e-5
    for name, a, b, c in zip(("out", "dq", "dk", "dv"), *res):
        err = float((a - c).abs().max()) / (float(c.abs().max()) + 1e-12)
        assert err < bound, (name, err)
        if name == "dq":
            assert float((a - b).abs().max()) / float(b.abs().max()) < 2e-5, name
        else:
            assert torch.equal(a, b), name
    out0 = Fh.SelfAttnFlashFn.apply(*(t.to(dev) for t in base), 4, 0.0, 77)
    assert float((res[0][0] - out0.cpu()).abs().max()) > 1e-3          # masks were applied


def test_flash_self_attention_keep_bits_are_the_counter_hash(dev):
    """The words the split-bf16 forward leaves for the backward (tatt_sattn_fwd_bits), decoded by the layout include/tatt_hip.h
    documents, against dropout_keep of csrc/common.h evaluated in integer torch arithmetic: every bit, two seeds."""
    from tatt_amd import ops
    B, Pn, h, site, pdrop = 2, 256, 2, 100, 0.1
    g = torch.Generator().manual_seed(8)
    Q, K, V = (torch.randn(B, Pn, 32 * h, generator=g).to(dev) for _ in range(3))
    M = 0xFFFFFFFF
    for seed in (0x1234567, 0x7FEDCBA987654321):
        sd = torch.tensor([seed], dtype=torch.int64, device=dev)
        ops.LIB.tatt_sattn_generation(2)
        O, lse = torch.empty_like(Q), torch.empty(B, h, Pn, device=dev)
        bits = torch.zeros(B * h * Pn * Pn // 32, device=dev, dtype=torch.int32)
        ops.call("tatt_sattn_fwd_bits", ops.P(Q), ops.P(K), ops.P(V), ops.P(O), ops.P(lse), ops.P(bits), B, Pn, h, 32 ** -0.5, pdrop, ops.P(sd),
                 site, ops.stream())
        k0 = (seed & M) ^ ((site * 0x9E3779B9) & M)
        k1 = ((seed >> 32) + site * 0x85EBCA77) & M
        x = torch.arange(B * h * Pn * Pn, device=dev, dtype=torch.int64) ^ k0
        x = ((x ^ (x >> 16)) * 0x85EBCA6B) & M
        x = (x + k1) & M
        x = ((x ^ (x >> 13)) * 0xC2B2AE35) & M
        x = x ^ (x >> 16)
        keep = (x >= int(pdrop * 4294967296.0)).view(B * h, Pn, Pn)
        nb = Pn // 32
        words = bits.view(B * h, nb, nb, 32).to(torch.int64) & M          # [bh][query block][key block][word]
        got = torch.zeros_like(keep)
        for d in range(32):
            v, half = d >> 1, d & 1
            key = (v & 3) + 8 * (v >> 2) + 4 * half
            for qq in range(32):
                got[:, qq::32, key::32] = ((words[:, :, :, d] >> qq) & 1).bool()
        assert torch.equal(got, keep), int((got != keep).sum())
        assert 0.88 < float(got.float().mean()) < 0.92


# ------------------------------------------------------------------------------------------- split-bf16 token projections
@pytest.mark.parametrize("N,K,K1,N1", [(192, 128, 64, 192), (192, 64, 64, 192), (128, 192, 192, 64), (64, 192, 192, 64), (64, 64, 64, 64),
                                        (64, 128, 128, 64)])
def test_tokgemm_split_bf16_vs_fp64(dev, N, K, K1, N1):
    """tatt_tokgemm_sb (Y = [X1 | X2] W^T + b on the bf16 matrix cores, hi/lo operand split) against fp64 for every instantiated
    (N, K): two sources / two destinations included; both packings (W (N, K) and the transposed view of a (K, N) matrix)."""
    from tatt_amd import ops
    from tatt_amd import functional as Fh
    M = 64 * 37
    X, W, b = R(M, K, seed=1), R(N, K, seed=2) / math.sqrt(K), R(N, seed=3)
    ref = X.double() @ W.double().t() + b.double()
    Xd = X.to(dev)
    for trans in (0, 1):
        src = (W if trans == 0 else W.t().contiguous()).to(dev)           # trans = 1: the operand is stored (K, N)
        Wpk = torch.empty(N * K, device=dev)
        ops.call("tatt_tokgemm_pack", ops.P(src), ops.P(Wpk), N, K, K if trans == 0 else N, trans, ops.stream())
        X1 = Xd[:, :K1].contiguous()
        X2 = Xd[:, K1:].contiguous() if K1 < K else None
        Y1, Y2 = Fh._tokgemm(X1, X2, Wpk, b.to(dev), N, K, N1)
        Y = Y1 if Y2 is None else torch.cat([Y1, Y2], 1)
        err = float((Y.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (trans, err)


@pytest.mark.parametrize("N,K", [(128, 128), (128, 64), (64, 128)])
def test_tokgemm_epilogues_relu_and_accumulate_vs_fp64(dev, N, K):
    """tatt_tokgemm_sb_ex: the (N, K) of the TBSRN FeatureEnhancer projections; ReLU after the bias; accumulation into the destination
    (the sum of data gradients dq Wq + dk Wk + dv Wv) -- against fp64."""
    from tatt_amd import ops
    from tatt_amd import functional as Fh
    M = 64 * 21
    X, W, b, Y0 = R(M, K, seed=21), R(N, K, seed=22) / math.sqrt(K), R(N, seed=23), R(M, N, seed=24)
    Wpk = torch.empty(N * K, device=dev)
    ops.call("tatt_tokgemm_pack", ops.P(W.to(dev)), ops.P(Wpk), N, K, K, 0, ops.stream())
    lin = X.double() @ W.double().t() + b.double()
    y = Fh._tokgemm_ex(X.to(dev), Wpk, b.to(dev), N, K, act=1)
    assert float((y.cpu().double() - lin.clamp_min(0)).abs().max() / lin.abs().max()) < 1e-5
    acc = Y0.clone().to(dev)
    Fh._tokgemm_ex(X.to(dev), Wpk, b.to(dev), N, K, out=acc, accum=True)
    assert float((acc.cpu().double() - (Y0.double() + lin)).abs().max() / lin.abs().max()) < 1e-5


def test_qkv_projection_operator_vs_three_linears(dev):
    """QKVProjFn (three prepacked split-bf16 projections of one token matrix; backward accumulates the three data gradients in the
    GEMM epilogues) against three fp64 linears: values, dx and the six parameter gradients; and LinearFn's prepacked path (ReLU)."""
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(4)
    B, Pn, E = 2, 256, 128
    x = torch.randn(B, Pn, E, generator=g)
    torch.manual_seed(41)                                              # (module initialisation: independent of the tests that ran before)
    lins = [torch.nn.Linear(E, E) for _ in range(3)] + [torch.nn.Linear(E, 64)]
    ws = [torch.randn(B, Pn, E, generator=g) for _ in range(3)]
    xd = x.double().requires_grad_(True)
    ref = [F.linear(xd, l.weight.detach().double(), l.bias.detach().double()) for l in lins[:3]]
    sum((r * w.double()).sum() for r, w in zip(ref, ws)).backward()
    ref_dx = xd.grad.clone()
    dl = [l.to(dev) for l in lins]
    Fh.linear_prepack(dl)
    try:
        xg = x.to(dev).requires_grad_(True)
        q, k, v = Fh.qkv_projection(xg, dl[0], dl[1], dl[2])
        assert isinstance(q.grad_fn, Fh.QKVProjFn._backward_cls)
        sum((o * w.to(dev)).sum() for o, w in zip((q, k, v), ws)).backward()
        for got, want in zip((q, k, v), ref):
            assert float((got.detach().cpu().double() - want.detach()).abs().max() / want.detach().abs().max()) < 1e-5
        assert float((xg.grad.cpu().double() - ref_dx).abs().max() / ref_dx.abs().max()) < 1e-5
        for l, w in zip(dl[:3], ws):
            want_w = (w.double().reshape(-1, E).t() @ x.double().reshape(-1, E))
            want_b = w.double().reshape(-1, E).sum(0)
            assert float((l.weight.grad.cpu().double() - want_w).abs().max() / want_w.abs().max()) < 1e-5
            assert float((l.bias.grad.cpu().double() - want_b).abs().max() / want_b.abs().max()) < 1e-5
        # LinearFn on a prepacked weight: 128 -> 64 with ReLU, forward and data gradient
        x2 = x.to(dev).requires_grad_(True)
        y = Fh.linear(x2, dl[3].weight, dl[3].bias, act=1)
        w4 = torch.randn(B, Pn, 64, generator=g)
        (y * w4.to(dev)).sum().backward()
        # (the float64 gradient takes the relu decisions of the GPU forward: a pre-activation within the arithmetic's error of zero --
        # 32,768 of them here -- may fall on the other side in float64, which moves that token's whole dx row by O(1))
        W4, b4 = lins[3].weight.detach().cpu().double(), lins[3].bias.detach().cpu().double()
        pre = F.linear(x.double(), W4, b4)
        kept = (y.detach().cpu() > 0).double()
        assert float((y.detach().cpu().double() - pre.clamp_min(0)).abs().max() / pre.abs().max()) < 1e-5
        assert float(((pre > 0).double() - kept).abs().sum()) <= 4                       # only decisions at |pre| ~ 1e-5 may differ
        want_dx = (w4.double() * kept) @ W4
        assert float((x2.grad.cpu().double() - want_dx).abs().max() / want_dx.abs().max()) < 1e-5
    finally:
        Fh.linear_prepack_done()


@pytest.mark.parametrize("M,N,K,splits", [(64 * 21, 128, 128, 128), (32, 128, 128, 128), (32 * 7, 64, 128, 3), (49152, 128, 128, 128),
                                            (32 * 50, 128, 64, 128), (32 * 9, 64, 64, 256)])
def test_tok_wgrad_split_bf16_vs_fp64(dev, M, N, K, splits):
    """tatt_tok_wgrad_sb + tatt_splitk_reduce: dW = dY^T X and db = column sums of dY in one pass over the tokens on the bf16 matrix
    cores (operands through the transposing LDS reads), against fp64: every instantiated (N, K), ragged chunk counts over the
    splits, a single chunk, the full-size token count; twice (bit-identical: the reduction order is fixed)."""
    from tatt_amd import ops
    dy, x = R(M, N, seed=31), R(M, K, seed=32)
    ref_w, ref_b = dy.double().t() @ x.double(), dy.double().sum(0)
    old = ops.TOK_WGRAD_SPLITS
    ops.TOK_WGRAD_SPLITS = splits
    try:
        outs = []
        for _ in range(2):
            dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            assert ops.tok_wgrad_takes(dy.to(dev), x.to(dev))
            ops.tok_wgrad_sb(dy.to(dev), x.to(dev), dW, db)
            outs.append((dW.cpu(), db.cpu()))
    finally:
        ops.TOK_WGRAD_SPLITS = old
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][0].double() - ref_w).abs().max() / ref_w.abs().max()) < 1e-5
    assert float((outs[0][1].double() - ref_b).abs().max() / ref_b.abs().max()) < 2e-6


def test_feed_forward_operator_equals_the_operator_chain(dev):
    """FeedForwardFn (dropout in the first GEMM's epilogue, relu' and dropout' in the epilogue of the second GEMM's data gradient) against
    linear -> dropout -> linear on the same prepacked weights: the same masks, so values and all gradients agree to round-off; and
    against fp64 without dropout."""
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(6)
    B, Pn, E = 2, 320, 128
    x = torch.randn(B, Pn, E, generator=g)
    w = torch.randn(B, Pn, E, generator=g)
    torch.manual_seed(42)
    l1, l2 = torch.nn.Linear(E, E).to(dev), torch.nn.Linear(E, E).to(dev)
    res = []
    for fused in (True, False):
        Fh.set_seed(dev, 9)
        Fh.begin_training_forward(dev)
        Fh.linear_prepack([l1, l2])
        Fh.FFN_FUSED = fused
        try:
            for l in (l1, l2):
                l.weight.grad = l.bias.grad = None
            xg = x.to(dev).requires_grad_(True)
            y = Fh.feed_forward(xg, l1, l2, 0.1, True, 41)
            assert isinstance(y.grad_fn, Fh.FeedForwardFn._backward_cls) == fused
            (y * w.to(dev)).sum().backward()
            res.append([t.detach().cpu().clone() for t in (y, xg.grad, l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad)])
        finally:
            Fh.FFN_FUSED = True
            Fh.linear_prepack_done()
    for name, a, b in zip(("y", "dx", "dw1", "db1", "dw2", "db2"), *res):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (name, float((a - b).abs().max()), float(b.abs().max()))
    assert float((res[0][0] == 0).float().mean()) < 0.01               # the dropout acts on the hidden layer, not on the output
    # no dropout (evaluation / p = 0): against fp64
    Fh.linear_prepack([l1, l2])
    try:
        xg = x.to(dev).requires_grad_(True)
        y = Fh.feed_forward(xg, l1, l2, 0.1, False, 41)
        (y * w.to(dev)).sum().backward()
    finally:
        Fh.linear_prepack_done()
    W1, b1, W2, b2 = (t.detach().cpu().double() for t in (l1.weight, l1.bias, l2.weight, l2.bias))
    pre = F.linear(x.double(), W1, b1)
    yr = F.linear(pre.clamp_min(0), W2, b2)
    assert float((y.detach().cpu().double() - yr).abs().max() / yr.abs().max()) < 2e-5
    # the float64 gradient with the relu decisions fixed to float64's own; a decision the split-bf16 forward takes differently (a
    # pre-activation within ~1e-5 of zero) moves one token's dx row by O(1): allow a handful of such rows, the rest to 2e-5
    want_dx = ((w.double() @ W2) * (pre > 0).double()) @ W1
    err = (xg.grad.cpu().double() - want_dx).abs().reshape(-1, E).max(1).values / want_dx.abs().max()
    assert int((err > 2e-5).sum()) <= 4 and float(err.median()) < 1e-5, (int((err > 2e-5).sum()), float(err.max()))


def test_feed_forward_layer_norm_operator_equals_the_two_operators(dev):
    """FeedForwardLnFn (the residual's gradient and w_1's data gradient summed in the GEMM epilogue, tatt_tokgemm_sb_add) against
    FeedForwardFn followed by LayerNormFn (autograd adds the two gradients of x): same masks and kernels, so values, dx and all six
    parameter gradients agree to the order of one addition."""
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(16)
    B, Pn, E = 2, 320, 128
    x = torch.randn(B, Pn, E, generator=g)
    w = torch.randn(B, Pn, E, generator=g)
    torch.manual_seed(42)
    l1, l2 = torch.nn.Linear(E, E).to(dev), torch.nn.Linear(E, E).to(dev)
    ga, be = (torch.rand(E, generator=g) + 0.5).to(dev).requires_grad_(True), torch.randn(E, generator=g).to(dev).requires_grad_(True)
    res = []
    for fused in (True, False):
        Fh.set_seed(dev, 9)
        Fh.begin_training_forward(dev)
        Fh.linear_prepack([l1, l2])
        Fh.FFN_LN_FUSED = fused
        try:
            for t in (l1.weight, l1.bias, l2.weight, l2.bias, ga, be):
                t.grad = None
            xg = x.to(dev).requires_grad_(True)
            y = Fh.feed_forward_ln(xg, l1, l2, ga, be, 1e-6, 1, 0.1, True, 43)
            assert isinstance(y.grad_fn, Fh.FeedForwardLnFn._backward_cls) == fused
            (y * w.to(dev)).sum().backward()
            res.append([t.detach().cpu().clone() for t in (y, xg.grad, l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad, ga.grad, be.grad)])
        finally:
            Fh.FFN_LN_FUSED = True
            Fh.linear_prepack_done()
    for name, a, b in zip(("y", "dx", "dw1", "db1", "dw2", "db2", "dgamma", "dbeta"), *res):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (name, float((a - b).abs().max()), float(b.abs().max()))


def test_attention_layer_norm_operator_equals_the_operator_chain(dev):
    """AttnLnFn (projections, score-free attention with dropout, output projection, residual LayerNorm; the residual's gradient is the
    addend of the first data-gradient GEMM into x) against the same launches as separate operators: values, dx and all ten parameter
    gradients to the order of one addition."""
    from tatt_amd import functional as Fh
    from tatt_amd.tbsrn import MultiHeadedAttention
    g = torch.Generator().manual_seed(26)
    B, Pn, E = 2, 256, 128
    x = torch.randn(B, Pn, E, generator=g)
    w = torch.randn(B, Pn, E, generator=g)
    torch.manual_seed(3)
    mh = MultiHeadedAttention(4, E).to(dev)
    for l in mh.linears:
        torch.nn.init.normal_(l.weight, std=0.08)
    ga, be = (torch.rand(E, generator=g) + 0.5).to(dev).requires_grad_(True), torch.randn(E, generator=g).to(dev).requires_grad_(True)
    leaves = [t for l in mh.linears for t in (l.weight, l.bias)] + [ga, be]
    res = []
    for fused in (True, False):
        Fh.set_seed(dev, 19)
        Fh.begin_training_forward(dev)
        Fh.linear_prepack(list(mh.linears))
        Fh.ATTN_LN_FUSED = fused
        try:
            for t in leaves:
                t.grad = None
            xg = x.to(dev).requires_grad_(True)
            y = Fh.attention_ln(xg, mh, ga, be, 1e-6, 1, 0.1, 57)
            assert isinstance(y.grad_fn, Fh.AttnLnFn._backward_cls) == fused
            (y * w.to(dev)).sum().backward()
            res.append([t.detach().cpu().clone() for t in [y, xg.grad] + [t.grad for t in leaves]])
        finally:
            Fh.ATTN_LN_FUSED = True
            Fh.linear_prepack_done()
    for i, (a, b) in enumerate(zip(*res)):
        assert float((a - b).abs().max()) <= 3e-6 * float(b.abs().max()), (i, float((a - b).abs().max()), float(b.abs().max()))


# ------------------------------------------------------------------------------------------- fused GruBlock weight gradients
@pytest.mark.parametrize("M,with_xb,groups", [(32 * 200, True, 128), (32 * 200, False, 128), (32, True, 128), (32 * 7, True, 3),
                                               (49152, True, 128), (49152, True, 256)])
def test_gru_wgrad_split_bf16_vs_fp64(dev, M, with_xb, groups):
    """tatt_gru_wgrad_sb + tatt_splitk_reduce: dW' = dgi^T [x | xb], db' = sum dgi, dW_hh = dgh^T hprev, db_hh = sum dgh of one
    GruBlock (reference model/tsrn.py:1075-1084) in one pass over the tokens on the bf16 matrix cores (hi/lo operand split), against
    fp64.  Ragged chunk counts over the persistent groups, a single chunk, no second input, and the full-size token count."""
    from tatt_amd import ops
    dgi, dgh = R(M, 192, seed=11), R(M, 192, seed=12)
    x, xb, hp = R(M, 64, seed=13), (R(M, 64, seed=14) if with_xb else None), R(M, 64, seed=15)
    K = 128 if with_xb else 64
    xx = torch.cat([x, xb], 1) if with_xb else x
    ref = (dgi.double().t() @ xx.double(), dgi.double().sum(0), dgh.double().t() @ hp.double(), dgh.double().sum(0))
    old = ops.GRU_WGRAD_GROUPS
    ops.GRU_WGRAD_GROUPS = groups
    try:
        outs = []
        for _ in range(2):
            dWp, dWhh = torch.empty(192, K, device=dev), torch.empty(192, 64, device=dev)
            dbp, dbhh = torch.empty(192, device=dev), torch.empty(192, device=dev)
            args = [t.to(dev) if t is not None else None for t in (dgi, dgh, x, xb, hp)]
            assert ops.gru_wgrad_fusable(*args)
            ops.gru_wgrad_sb(*args, dWp, dWhh, dbp, dbhh)
            outs.append((dWp, dbp, dWhh, dbhh))
    finally:
        ops.GRU_WGRAD_GROUPS = old
    for name, got, want in zip(("dWp", "dbp", "dWhh", "dbhh"), outs[0], ref):
        err = float((got.cpu().double() - want).abs().max() / want.abs().max())
        assert err < 2e-5, (name, err)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                                  # deterministic


@pytest.mark.parametrize("M,N,K,split", [(3072, 1536, 512, 6), (3072, 1536, 512, 4), (3072, 1536, 512, 96), (128, 384, 128, 6),
                                         (320, 1536, 512, 6), (32, 128, 256, 6)])
def test_query_gru_recurrent_weight_gradient_split_bf16_vs_fp64(dev, M, N, K, split):
    """tatt_qgru_wgrad_sb + tatt_splitk_reduce: dW_hh = dgh^T h_prev and db_hh = sum dgh for both directions of the query GRU in one
    launch (the gradient nn.GRU accumulates over its time steps, reference model/transformer_v2.py:201-221) against fp64: the
    published geometry (48 steps x 64 sequences, 3 x 512 gates), ragged split counts (10 chunks over 6 -> 5 splits), one chunk."""
    from tatt_amd import ops
    A = [R(M, N, seed=21 + d) for d in range(2)]
    B = [R(M, K, seed=31 + d) for d in range(2)]
    ref = []
    for d in range(2):
        ref += [A[d].double().t() @ B[d].double(), A[d].double().sum(0)]
    Ad, Bd = [a.to(dev) for a in A], [b.to(dev) for b in B]
    assert ops.qgru_wgrad_takes(Ad[0], Bd[0])
    assert not ops.qgru_wgrad_takes(Ad[0][:, :100].contiguous(), Bd[0])
    old = ops.QGRU_WGRAD_SPLIT
    ops.QGRU_WGRAD_SPLIT = split
    try:
        outs = [ops.qgru_wgrad_sb(Ad[0], Ad[1], Bd[0], Bd[1]) for _ in range(2)]
    finally:
        ops.QGRU_WGRAD_SPLIT = old
    for name, got, want in zip(("dW0", "db0", "dW1", "db1"), outs[0], ref):
        err = float((got.cpu().double() - want).abs().max() / want.abs().max())
        assert err < 2e-5, (name, err)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                                  # deterministic


# ------------------------------------------------------------------------------------------- split-bf16 3x3 weight gradient
@pytest.mark.parametrize("gen", [2, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 16, 64, 64, 64), (2, 8, 128, 64, 256), (1, 4, 64, 256, 64), (5, 3, 192, 64, 64), (48, 16, 64, 64, 64),
                                            (1, 4, 16, 64, 64), (7, 8, 48, 64, 64)])
def test_conv3_wgrad_split_bf16_vs_fp64(dev, B, H, W, Cin, Cout, gen):
    """tatt_conv3_c64_wgrad_partial_sb (pixels as the contraction axis) against autograd in fp64 and against the fp32-MFMA kernel, both
    generations: 2 = 4 x 16-pixel tiles, operands through gfx950's transposing LDS read, staging waves beside MFMA waves, bias gradient
    summed in fp32 by the staging waves (round 6); 1 = row segments, operands transposed through an LDS fragment buffer, bias gradient from
    an all-ones A fragment (H = 3 and anything else generation 2 does not tile runs it whatever the switch says).  Channel blocks on both
    sides, heights that leave halo rows outside the image, the benchmark shape (6 tiles per work-group), a single tile, a ragged tile
    count (more groups than some groups have tiles)."""
    from tatt_amd import ops
    from tatt_amd._lib import LIB
    if gen == 1 and W % 64:
        pytest.skip("the row-segment kernel walks 64-pixel segments")
    LIB.tatt_conv3_wgrad_sb_generation(gen)
    ops.CONV3_SB_NARROW_MAPS = True
    try:
        _conv3_wgrad_case(dev, B, H, W, Cin, Cout)
    finally:
        LIB.tatt_conv3_wgrad_sb_generation(2)
        ops.CONV3_SB_NARROW_MAPS = False


def _conv3_wgrad_case(dev, B, H, W, Cin, Cout):
    from tatt_amd import ops
    g = torch.Generator().manual_seed(61)
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, H, W, Cout, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w, b, padding=1).backward(dy.permute(0, 3, 1, 2).double())
    xd, dyd = x.to(dev), dy.to(dev)
    assert ops.CONV3_WGRAD_SB
    dw, db = ops.conv_wgrad(xd, dyd, Cout, 3, 3, want_db=True)
    err = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
    assert err < 2e-5, err
    errb = float((db.double().cpu() - b.grad).abs().max() / b.grad.abs().max())
    assert errb < 2e-5, errb
    assert torch.equal(dw, ops.conv_wgrad(xd, dyd, Cout, 3, 3))                      # deterministic, with or without the bias part
    ops.CONV3_WGRAD_SB = False
    try:
        dw32 = ops.conv_wgrad(xd, dyd, Cout, 3, 3) if W % 64 == 0 else None
    finally:
        ops.CONV3_WGRAD_SB = True
    if W % 64 == 0:
        assert float((dw - dw32).abs().max() / dw32.abs().max()) < 3e-5


# ------------------------------------------------------------------------------------------- second-generation BiGRU recurrences
def _gru32_inputs(B, H, W, seed):
    M = B * H * W
    whh = [R(96, 32, seed=seed + k, scale=0.3) for k in range(2)]
    bhh = [R(96, seed=seed + 2 + k, scale=0.3) for k in range(2)]
    return R(M, 192, seed=seed + 4), whh, bhh, R(M, 64, seed=seed + 5)


@pytest.mark.parametrize("vertical", [True, False])
@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (3, 5, 7), (1, 8, 12), (1, 1, 3)])
def test_gru32_v2_matches_v1(dev, vertical, B, H, W):
    """tatt_gru32_fwd2 / tatt_gru32_bwd2 (one wave per (sequence, direction), halves joined by v_permlane32_swap; model/tsrn.py:1072)
    against the first-generation kernels on the same inputs: same outputs up to the order of fp32 summation.  Odd sequence counts
    (padding waves), T not a multiple of the prefetch depth, T = 1."""
    from tatt_amd import ops
    gi, whh, bhh, dout = _gru32_inputs(B, H, W, 40)
    geom = ops.seq_geom(B, H, W, vertical)
    d = lambda t: t.to(dev)

    def run(v2, saved=None):
        ops.GRU32_V2, min_t = v2, ops.GRU32_FWD_V2_MIN_T
        ops.GRU32_FWD_V2_MIN_T = 0                                  # the second-generation forward at every length
        try:
            out, gates = ops.gru32_fwd(d(gi), d(whh[0]), d(bhh[0]), d(whh[1]), d(bhh[1]), geom, save=True)
            out_ns, none = ops.gru32_fwd(d(gi), d(whh[0]), d(bhh[0]), d(whh[1]), d(bhh[1]), geom, save=False)
            assert none is None and torch.equal(out, out_ns)
            o, g = saved if saved is not None else (out, gates)      # both backward generations start from the SAME saved forward
            return (out, gates) + tuple(ops.gru32_bwd(g, o, d(dout), d(whh[0]), d(whh[1]), geom))
        finally:
            ops.GRU32_V2, ops.GRU32_FWD_V2_MIN_T = True, min_t

    r1 = run(False)
    r2 = run(True, saved=r1[:2])
    for name, a, b in zip(("out", "gates", "dgi", "dgh", "hprev"), r1, r2):
        err = float((a - b).abs().max() / (a.abs().max() + 1e-12))
        assert err < 5e-6, (name, err)


@pytest.mark.parametrize("vertical,B,H,W,cat,groups", [(True, 2, 16, 64, True, 128), (False, 2, 16, 64, False, 128), (False, 1, 8, 32, True, 3),
                                                       (True, 3, 8, 16, True, 1), (True, 48, 16, 64, True, 128), (False, 48, 16, 64, False, 256)])
def test_gru_wgrad_frag_vs_fp64(dev, vertical, B, H, W, cat, groups):
    """tatt_gru32_bwd2 with fragment emission + tatt_gru_wgrad_frag + reductions: dW' = dgi^T [x | xb], db' = sum dgi, dW_hh (compact:
    [forward; reverse]) = the diagonal blocks of dgh^T hprev, db_hh = sum dgh of one GruBlock (reference model/tsrn.py:1075-1084)
    against fp64 products of the dgi / dgh / hprev the same recurrence writes without fragments; dgi itself must not change."""
    from tatt_amd import ops
    gi, whh, bhh, dout = _gru32_inputs(B, H, W, 50)
    geom = ops.seq_geom(B, H, W, vertical)
    assert ops.gru_frag_ok(geom)
    M = B * H * W
    x, xb = R(M, 64, seed=61), (R(M, 64, seed=62) if cat else None)
    K = 128 if cat else 64
    d = lambda t: None if t is None else t.to(dev)
    out, gates = ops.gru32_fwd(d(gi), d(whh[0]), d(bhh[0]), d(whh[1]), d(bhh[1]), geom, save=True)
    dgi, dgh, hprev = ops.gru32_bwd(gates, out, d(dout), d(whh[0]), d(whh[1]), geom)
    dgi64, dgh64, hp64 = dgi.cpu().double(), dgh.cpu().double(), hprev.cpu().double()
    xx = (torch.cat([x, xb], 1) if cat else x).double()
    full = dgh64.t() @ hp64
    ref = (dgi64.t() @ xx, dgi64.sum(0), torch.cat([full[:96, :32], full[96:, 32:]], 0), dgh64.sum(0))
    old = ops.GRU_WGRAD_FRAG_GROUPS
    ops.GRU_WGRAD_FRAG_GROUPS = groups
    try:
        outs = []
        for _ in range(2):
            dgi2, frag = ops.gru32_bwd_frag(gates, out, d(dout), d(whh[0]), d(whh[1]), geom)
            assert torch.equal(dgi2, dgi)
            dWp, dWhh = torch.empty(192, K, device=dev), torch.empty(192, 32, device=dev)
            dbp, dbhh = torch.empty(192, device=dev), torch.empty(192, device=dev)
            ops.gru_wgrad_frag(frag, d(x), d(xb), geom, dWp, dWhh, dbp, dbhh)
            outs.append((dWp, dbp, dWhh, dbhh))
    finally:
        ops.GRU_WGRAD_FRAG_GROUPS = old
    for name, got, want in zip(("dWp", "dbp", "dWhh", "dbhh"), outs[0], ref):
        err = float((got.cpu().double() - want).abs().max() / want.abs().max())
        assert err < 2e-5, (name, err)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                                  # deterministic
