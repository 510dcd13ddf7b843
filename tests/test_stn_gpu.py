"""The STN head as a few launches (csrc/stnhead.hip, tatt_amd.functional.StnHeadFn) against
(a) the operator-by-operator HIP path it replaces (tsrn.STN_FUSED = False) and
(b) the same head evaluated by torch in float64 on the CPU (the modules of the parameter holder are ordinary nn.Modules; reference
    model/stn_head.py:25-106): control points, running statistics, num_batches_tracked and EVERY parameter gradient.
Below train-mode BatchNorms with ReLU / max-pool selections either fp32 path sits some distance from fp64 (selections flip on
round-off); the fused path is held to the operator chain's own distance.  The launches synchronise their work-groups in flight
(write-through partial sums + flag words): the repeat test runs them beside a second busy stream and compares every word."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _head(seed=4):
    import tatt_amd.tsrn as T
    torch.manual_seed(seed)
    stn = T.STNHead(4, 20, "none")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in stn.named_parameters():
            if p.dim() == 1:                                   # non-trivial BatchNorm affines and biases
                p.add_(0.2 * torch.randn(p.shape, generator=g))
        stn.stn_fc2.weight.add_(0.05 * torch.randn(stn.stn_fc2.weight.shape, generator=g))     # zero-initialised in the reference
        stn.stn_fc1[0].weight.mul_(30.0)                       # N(0, 0.001) init: lift it so that gradients below are not denormal-small
    return stn.train()


def _inputs(B, seed=9):
    g = torch.Generator().manual_seed(seed + B)
    x = torch.rand(B, 4, 16, 64, generator=g)
    x[:, 3] = (x[:, 3] > 0.5).float()
    dctrl = torch.randn(B, 20, 2, generator=g)
    return x, dctrl


def _run_hip(stn0, x, dctrl, fused, dev):
    import tatt_amd.tsrn as T
    from tatt_amd import functional as Fh
    stn = copy.deepcopy(stn0).to(dev)
    old = T.STN_FUSED
    T.STN_FUSED = fused
    try:
        ctrl = T._stn_forward(x.to(dev), stn, True)
        ctrl.backward(dctrl.to(dev))
        torch.cuda.synchronize()
        Fh.sync_check()
    finally:
        T.STN_FUSED = old
    out = {"ctrl": ctrl.detach().cpu()}
    for n, p in stn.named_parameters():
        out["d." + n] = p.grad.detach().cpu()
    for n, b in stn.named_buffers():
        out["b." + n] = b.detach().cpu()
    return out


def _run_fp64(stn0, x, dctrl):
    stn = copy.deepcopy(stn0).double()
    B = x.shape[0]
    h = stn.stn_convnet(x.double())
    h = stn.stn_fc1(h.view(B, -1))
    ctrl = stn.stn_fc2(0.1 * h).view(B, 20, 2)
    ctrl.backward(dctrl.double())
    out = {"ctrl": ctrl.detach()}
    for n, p in stn.named_parameters():
        out["d." + n] = p.grad.detach()
    for n, b in stn.named_buffers():
        out["b." + n] = b.detach()
    return out


@pytest.mark.parametrize("B", [2, 5, 48, 64])
def test_stn_head_fused_vs_operator_chain_and_fp64(dev, B):
    stn0 = _head()
    x, dctrl = _inputs(B)
    ref = _run_fp64(stn0, x, dctrl)
    chain = _run_hip(stn0, x, dctrl, False, dev)
    fused = _run_hip(stn0, x, dctrl, True, dev)
    assert set(fused) == set(chain) == set(ref)
    worst = (0.0, None)
    for k in sorted(ref):
        if k.endswith("num_batches_tracked"):
            assert int(fused[k]) == int(chain[k]) == int(ref[k]) == 1, k
            continue
        scale = float(ref[k].abs().max()) + 1e-30
        e_chain = float((chain[k].double() - ref[k]).abs().max()) / scale
        e_fused = float((fused[k].double() - ref[k]).abs().max()) / scale
        is_conv_bias = k.startswith("d.stn_convnet") and k.endswith(".0.bias") or k == "d.stn_fc1.0.bias"
        if is_conv_bias:
            # a bias in front of a train-mode BatchNorm: the gradient is mathematically zero, every implementation returns round-off
            lim = 1e-4 * float(ref["d." + k[2:].replace("bias", "weight")].abs().max()) * (x.shape[0] * 16 * 64) ** 0.5
            assert float(fused[k].abs().max()) <= lim, (k, float(fused[k].abs().max()), lim)
            continue
        worst = max(worst, (e_fused, k))
        assert e_fused <= 4.0 * e_chain + 2e-5, (k, e_fused, e_chain)
    print("B = %d: worst fused-vs-fp64 %.2e (%s)" % (B, worst[0], worst[1]))


@pytest.mark.parametrize("B", [5, 48])
def test_stn_head_split_sums_folded_into_the_batchnorm_launches(dev, B):
    """The head's deep convolutions (forward and data gradient) split their contraction over the CUs; by default the partial maps are
    summed by the BatchNorm launch that consumes them (tatt_conv2d_fwd_partials -> tatt_stn_bn_pool_{fwd,bwd}_parts) instead of a
    tatt_splitk_reduce launch of their own.  Same terms, different order of addition: both forms sit at the same distance from fp64."""
    from tatt_amd import functional as Fh
    stn0 = _head()
    x, dctrl = _inputs(B)
    ref = _run_fp64(stn0, x, dctrl)
    assert Fh.STN_FOLD_SPLITS
    folded = _run_hip(stn0, x, dctrl, True, dev)
    Fh.STN_FOLD_SPLITS = False
    try:
        separate = _run_hip(stn0, x, dctrl, True, dev)
    finally:
        Fh.STN_FOLD_SPLITS = True
    for k in sorted(ref):
        if k.endswith("num_batches_tracked") or (k.startswith("d.stn_convnet") and k.endswith(".0.bias")) or k == "d.stn_fc1.0.bias":
            continue
        scale = float(ref[k].abs().max()) + 1e-30
        e_f = float((folded[k].double() - ref[k]).abs().max()) / scale
        e_s = float((separate[k].double() - ref[k]).abs().max()) / scale
        assert e_f <= 4.0 * e_s + 2e-5 and e_s <= 4.0 * e_f + 2e-5, (k, e_f, e_s)


def test_stn_head_fused_is_deterministic_under_load(dev):
    stn0 = _head()
    x, dctrl = _inputs(48)
    ref = _run_hip(stn0, x, dctrl, True, dev)
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device=dev)
    for rep in range(25):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 4):
                junk.mul_(1.0001)
        out = _run_hip(stn0, x, dctrl, True, dev)
        for k in ref:
            assert torch.equal(ref[k], out[k]), (rep, k)
    torch.cuda.synchronize()


@pytest.mark.parametrize("groups,lds", [(32, 0), (64, 65536)])
def test_stn_head_fused_beside_a_cu_holder(dev, groups, lds):
    """The fused head's launches (<= 128 work-groups exchanging BatchNorm partial sums in flight) beside resident work-groups that hold
    CUs for ~1 ms on a second stream -- what a collective's channel kernels do: bit-identical to the unloaded run, no expired wait."""
    from tatt_amd import ops, functional as Fh
    stn0 = _head()
    x, dctrl = _inputs(48)
    ref = _run_hip(stn0, x, dctrl, True, dev)
    sticky = Fh.sticky_word(dev)
    side = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    for rep in range(6):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            ops.call("tatt_cu_holder", groups, 100000, lds, ops.P(sink), ops.stream())
        out = _run_hip(stn0, x, dctrl, True, dev)
        for k in ref:
            assert torch.equal(ref[k], out[k]), (rep, k)
    torch.cuda.synchronize()
    Fh.sync_check()
    assert int(sticky[0].item()) == 0


def test_sync_guard_is_silent_on_a_clean_device_and_capacity_covers_the_grids(dev):
    """tatt_sync_guard launches and returns on a clean sticky word (a raised word would kill the process: not exercised here); the
    occupancy query says the whole MI355X holds the grids of every in-flight-synchronising launch."""
    from tatt_amd import ops, functional as Fh
    Fh.sticky_word(dev)
    ops.call("tatt_sync_guard", ops.stream())
    torch.cuda.synchronize()
    cap = Fh.sync_capacity(dev)
    assert min(cap[:4]) >= 256 and cap[4] >= 128 and cap[5] >= 32, cap


def test_stn_head_falls_back_for_other_geometries(dev):
    """B > 64 (the fully connected launch holds <= 64 samples) runs operator by operator, silently and correctly."""
    import tatt_amd.tsrn as T
    stn0 = _head()
    x, dctrl = _inputs(70)
    a = _run_hip(stn0, x, dctrl, True, dev)
    b = _run_hip(stn0, x, dctrl, False, dev)
    for k in a:
        assert torch.equal(a[k], b[k]), k
