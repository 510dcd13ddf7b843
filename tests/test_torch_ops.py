"""The operator-registry view of the HIP kernels (tatt_amd/torch_ops.py, `torch.ops.tatt_hip.*`; BASELINE.json north_star: "bound as custom
PyTorch ops").  Without a GPU: every op is registered with a schema and a fake (meta) kernel, shapes propagate under FakeTensorMode, and a
CPU tensor is refused (there is no CPU implementation).  On the GPU: each op equals the product's own call of the same kernel, autograd
through the registered formulas equals the product's autograd.Function, and torch.library.opcheck accepts the registrations."""
import pytest
import torch

import tatt_amd.torch_ops as T


def test_ops_are_registered_with_fake_kernels():
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in T.OPS:
        assert hasattr(torch.ops.tatt_hip, name), name
    with FakeTensorMode():
        cu = lambda *s: torch.empty(*s, device="cuda")
        x, w, b = cu(2, 16, 64, 64), cu(64, 64, 3, 3), cu(64)
        assert tuple(torch.ops.tatt_hip.conv2d(x, w, b, 0).shape) == (2, 16, 64, 64)
        assert tuple(torch.ops.tatt_hip.conv2d(cu(2, 32, 128, 64), cu(4, 64, 9, 9), cu(4), 0).shape) == (2, 32, 128, 4)
        assert tuple(torch.ops.tatt_hip.conv2d_dgrad(x, cu(64, 4, 9, 9)).shape) == (2, 16, 64, 4)
        dw, db = torch.ops.tatt_hip.conv2d_wgrad(x, x, 3, 3)
        assert tuple(dw.shape) == (64, 64, 3, 3) and tuple(db.shape) == (64,)
        out, gates = torch.ops.tatt_hip.gru32_fwd(cu(2048, 192), cu(96, 32), cu(96), cu(96, 32), cu(96), 2, 16, 64, True)
        assert tuple(out.shape) == (2048, 64) and tuple(gates.shape) == (2048, 256)
        dgi, dgh, hp = torch.ops.tatt_hip.gru32_bwd(gates, out, out, cu(96, 32), cu(96, 32), 2, 16, 64, True)
        assert tuple(dgi.shape) == (2048, 192) and tuple(hp.shape) == (2048, 64)
        y, m, r = torch.ops.tatt_hip.bn_train(cu(100, 64), cu(64), cu(64), cu(64), cu(64), 0.1, 1e-5, 2)
        assert tuple(y.shape) == (100, 64) and tuple(m.shape) == (64,)
        ctx_, wts = torch.ops.tatt_hip.attn_core(cu(2, 1024, 64), cu(2, 26, 64), cu(2, 26, 64))
        assert tuple(wts.shape) == (2, 1024, 26)
        assert tuple(torch.ops.tatt_hip.grid_sample(cu(2, 4, 16, 64), cu(2, 1024, 2)).shape) == (2, 16, 64, 4)
        assert tuple(torch.ops.tatt_hip.image_loss(cu(3, 4, 32, 128), cu(3, 4, 32, 128), 1.0, 1e-4).shape) == (3,)


def test_ops_refuse_cpu_tensors():
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.tatt_hip.conv2d(torch.zeros(1, 4, 64, 64), torch.zeros(64, 64, 3, 3), None, 0)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.tatt_hip.linear(torch.zeros(8, 64), torch.zeros(64, 64), None, 0)


@pytest.mark.gpu
def test_registry_ops_equal_the_product_path(dev):
    from tatt_amd import functional as Fh, ops
    g = torch.Generator().manual_seed(3)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    # convolution: forward and autograd through the registered formulas == the product's Conv2dFn
    for (cin, cout, k) in ((64, 64, 3), (4, 64, 9), (64, 4, 9)):
        x, w, b = R(2, 16, 64, cin), R(cout, cin, k, k) * 0.05, R(cout)
        res = []
        for fn in (lambda x, w, b: torch.ops.tatt_hip.conv2d(x, w, b, 0), lambda x, w, b: Fh.conv2d(x, w, b)):
            xs, ws, bs = (t.clone().requires_grad_(True) for t in (x, w, b))
            y = fn(xs, ws, bs)
            (y * y).sum().backward()
            res.append((y.detach(), xs.grad, ws.grad, bs.grad))
        for a, c in zip(*res):
            assert torch.equal(a, c)
    # linear (+ ReLU)
    x, w, b = R(300, 64), R(96, 64) * 0.1, R(96)
    res = []
    for fn in (lambda x, w, b: torch.ops.tatt_hip.linear(x, w, b, 1), lambda x, w, b: Fh.linear(x, w, b, act=1)):
        xs, ws, bs = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = fn(xs, ws, bs)
        (y * y).sum().backward()
        res.append((y.detach(), xs.grad, ws.grad, bs.grad))
    for a, c in zip(*res):
        assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max())
    # recurrences, BatchNorm, attention core, sampler, loss: the same kernels as the product's wrappers
    B, H, W = 2, 16, 64
    gi, whh, bhh, dout = R(B * H * W, 192), R(96, 32) * 0.3, R(96) * 0.3, R(B * H * W, 64)
    out, gates = torch.ops.tatt_hip.gru32_fwd(gi, whh, bhh, whh, bhh, B, H, W, False)
    o2, g2 = ops.gru32_fwd(gi, whh, bhh, whh, bhh, ops.seq_geom(B, H, W, False), save=True)
    assert torch.equal(out, o2) and torch.equal(gates, g2)
    for a, c in zip(torch.ops.tatt_hip.gru32_bwd(gates, out, dout, whh, whh, B, H, W, False),
                    ops.gru32_bwd(gates, out, dout, whh, whh, ops.seq_geom(B, H, W, False))):
        assert torch.equal(a, c)
    x2, ga, be = R(500, 64), R(64) * 0.2 + 1, R(64) * 0.2
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    y, mean, rstd = torch.ops.tatt_hip.bn_train(x2, ga, be, rm, rv, 0.1, 1e-5, 2)
    ref = torch.nn.functional.batch_norm(x2, None, None, ga, be, True, 0.1, 1e-5)
    ref = ref * torch.tanh(torch.nn.functional.softplus(ref))
    assert float((y - ref).abs().max()) < 1e-5 and float(rm.abs().max()) > 0
    dx, dg, db = torch.ops.tatt_hip.bn_backward(x2, R(500, 64), mean, rstd, ga, be, 2)
    assert dx.shape == x2.shape and torch.isfinite(dx).all()
    q, k, v = R(2, 70, 64) * 0.25, R(2, 26, 64), R(2, 26, 64)
    ctx_, wts = torch.ops.tatt_hip.attn_core(q, k, v)
    qh, kh, vh = (t.reshape(2, -1, 4, 16).permute(0, 2, 1, 3) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2), -1)
    assert float((ctx_ - (p @ vh).permute(0, 2, 1, 3).reshape(2, 70, 64)).abs().max()) < 1e-5
    assert float((wts - p.mean(1)).abs().max()) < 1e-6
    sr, hr = torch.rand(3, 4, 32, 128, generator=g).to(dev), torch.rand(3, 4, 32, 128, generator=g).to(dev)
    per = torch.ops.tatt_hip.image_loss(sr, hr, 1.0, 1e-4)
    assert torch.equal(per, Fh.ImageLossFn.apply(sr, hr, 1.0, 1e-4, None))


@pytest.mark.gpu
def test_opcheck_accepts_the_registrations(dev):
    g = torch.Generator().manual_seed(4)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    x, w, b = R(1, 8, 64, 64).requires_grad_(True), (R(64, 64, 3, 3) * 0.05).requires_grad_(True), R(64).requires_grad_(True)
    torch.library.opcheck(torch.ops.tatt_hip.conv2d.default, (x, w, b, 0), test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    gi, whh, bhh = R(512, 192), R(96, 32) * 0.3, R(96) * 0.3
    torch.library.opcheck(torch.ops.tatt_hip.gru32_fwd.default, (gi, whh, bhh, whh, bhh, 1, 8, 64, True), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.tatt_hip.bn_train.default,
                          (R(200, 64), R(64), R(64), torch.zeros(64, device=dev), torch.ones(64, device=dev), 0.1, 1e-5, 0),
                          test_utils=("test_schema", "test_faketensor"))
