import numpy as np
import torch


def to_dev(t, dev, grad=False):
    return t.detach().clone().to(dev).requires_grad_(grad)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def max_err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def check_close(name, got, ref, rtol=2e-4, atol=2e-5):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bool(bad.any()):
        i = int(torch.argmax(err - tol))
        raise AssertionError("%s: %d/%d elements off; worst |d|=%.3e (got %.6g ref %.6g), rel-norm err %.3e" % (
            name, int(bad.sum()), bad.numel(), float(err.reshape(-1)[i]), float(got.reshape(-1)[i]),
            float(ref.reshape(-1)[i]), rel_err(got, ref)))


def compare_fn(name, hip_fn, ref_fn, inputs, dev, grad_mask=None, rtol=2e-4, atol=2e-5, grtol=5e-4, gatol=5e-5):
    """Run hip_fn on GPU copies and ref_fn on CPU copies of `inputs` (list of tensors / None); compare
    outputs (tuple or tensor) and gradients of a random linear functional of the outputs."""
    grad_mask = grad_mask if grad_mask is not None else [t is not None and t.is_floating_point() for t in inputs]
    cin = [None if t is None else t.detach().clone().requires_grad_(bool(g)) for t, g in zip(inputs, grad_mask)]
    gin = [None if t is None else t.detach().clone().to(dev).requires_grad_(bool(g)) for t, g in zip(inputs, grad_mask)]
    ro = ref_fn(*cin)
    go = hip_fn(*gin)
    ro = ro if isinstance(ro, (tuple, list)) else (ro,)
    go = go if isinstance(go, (tuple, list)) else (go,)
    assert len(ro) == len(go)
    gen = torch.Generator().manual_seed(123)
    lr = lg = 0.0
    for k, (r, g) in enumerate(zip(ro, go)):
        check_close("%s.out%d" % (name, k), g, r, rtol, atol)
        w = torch.randn(r.shape, generator=gen)
        lr = lr + (r * w).sum()
        lg = lg + (g * w.to(dev)).sum()
    if any(grad_mask):
        lr.backward()
        lg.backward()
        for k, (c, g, m) in enumerate(zip(cin, gin, grad_mask)):
            if m:
                assert g.grad is not None, "%s: no grad for input %d" % (name, k)
                check_close("%s.grad%d" % (name, k), g.grad, c.grad, grtol, gatol * max(1.0, float(c.grad.abs().max())))


import re

# conv biases that feed a BatchNorm (train mode): their gradient is mathematically zero -- whatever an implementation
# returns is round-off noise (and Adam's first step turns that noise into +-lr).  Compared as "both negligible".
STRUCTURAL_ZERO_GRAD = re.compile(r"^(block[2-6]\.conv[12]\.bias|block7\.0\.bias|stn_head\.stn_convnet\.\d+\.0\.bias|"
                                  r"stn_head\.stn_fc1\.0\.bias)$")


def compare_param_grads(named_params, oracle_grads, rtol, rtol_stn=None):
    """-> (worst key, worst relative error); asserts structure (None grads, structural zeros)."""
    scale = max(float(g.abs().max()) for g in oracle_grads.values() if g is not None)
    worst = ("", 0.0)
    for k, p in named_params:
        og = oracle_grads[k]
        if og is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        pg = p.grad.detach().cpu()
        if STRUCTURAL_ZERO_GRAD.match(k):
            assert float(pg.abs().max()) < 1e-4 * scale and float(og.abs().max()) < 1e-4 * scale, k
            continue
        r = float((pg - og).norm() / (og.norm() + 1e-7 * scale * og.numel() ** 0.5))
        lim = rtol_stn if (rtol_stn is not None and k.startswith("stn_head")) else rtol
        if og.numel() == 1 and rtol_stn is not None:
            lim = 5e-2       # single-slope PReLU gradients are cancelling sums; with the STN's coordinate noise upstream ~1e-2
        assert r < lim, (k, r, lim)
        if r > worst[1]:
            worst = (k, r)
    return worst


class TorchStepKernels:
    """TEST INFRASTRUCTURE: torch stand-ins for tatt_l2norm / tatt_adam_step (same argument meaning), injected into
    tatt_amd.train.Trainer by the CPU (gloo) tests of its data-parallel orchestration.  Formulas = oracle.clip_grad_norm /
    oracle.adam_step; the product's own default (HipStepKernels) has no CPU path."""

    def l2norm(self, g, out, ws):
        out.copy_(torch.sqrt((g.double() ** 2).sum()).float().reshape(1))

    def adam(self, p, g, m, v, lr, b1, b2, eps, gnorm, max_norm, gscale, step):
        from oracle import tatt_oracle as O
        coef = gscale
        if max_norm > 0.0:
            coef = coef * min(1.0, max_norm / (float(gnorm) * gscale + 1e-6))
        p1, m1, v1 = O.adam_step(p, g * coef, m, v, int(step), lr, (b1, b2), eps)
        p.copy_(p1), m.copy_(m1), v.copy_(v1)
