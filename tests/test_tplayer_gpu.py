"""The one-kernel transformer layer of the TP interpreter (csrc/tplayer.hip, tatt_amd.functional.TPStackFn) against
(a) the operator-by-operator HIP path it replaces -- with dropout ON the two draw the same masks (seed word, site, flat index),
    so outputs and every gradient must agree to accumulation-order round-off -- and
(b) the CPU oracle (reference model/transformer_v2.py:470-484, 806-833, 380-390; model/tsrn.py:194-224) with dropout off.
Covers the benchmark geometry (P = 1024 queries, 26 keys), ragged tiles (P not a multiple of 32), one decoder layer, the
gradient through the returned attention weights, and the encoder geometry (26 queries of the sample's own memory)."""
import pytest
import torch

from oracle import tatt_oracle as O
from tests.util import max_err

pytestmark = pytest.mark.gpu


def _interp(H, W, n_dec, seed=3):
    import tatt_amd.tsrn as T
    torch.manual_seed(seed)
    ig = T.TPInterpreter(37, 64, output_size=(H, W), t_decoder_num=n_dec)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in ig.named_parameters():          # non-trivial LayerNorm affines and biases (default init: 1 / 0)
            if p.dim() == 1 and p.numel() > 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    return ig


def _run(ig, feat, tp, qpos, fused, drop, dev, use_w=True, seed=77):
    """-> dict of outputs and gradients (CPU tensors) of one forward + backward of the interpreter's transformer."""
    import tatt_amd.tsrn as T
    from tatt_amd import functional as Fh
    T.TP_FUSED = fused
    ig.dropout_on = drop
    for p in ig.parameters():
        p.grad = None
    f = feat.detach().clone().to(dev).requires_grad_(True)
    t = tp.detach().clone().to(dev).requires_grad_(True)
    q = qpos.detach().clone().to(dev).requires_grad_(True)
    Fh.set_seed(dev, seed)
    Fh.begin_training_forward(dev)
    try:
        tp_map, wts = T._tp_interpreter(f, t, ig, True, qpos=q)
    finally:
        T.TP_FUSED = True
    g = torch.Generator().manual_seed(5)
    w1 = torch.randn(tp_map.shape, generator=g).to(dev)
    w2 = torch.randn(wts.shape, generator=g).to(dev)
    loss = (tp_map * w1).sum() + ((wts * w2).sum() * 3.0 if use_w else 0.0)
    loss.backward()
    out = {"tp_map": tp_map, "wts": wts, "d.feat": f.grad, "d.tp": t.grad, "d.qpos": q.grad}
    for n, p in ig.named_parameters():
        out["g." + n] = p.grad
    return {k: (None if v is None else v.detach().float().cpu().clone()) for k, v in out.items()}


def _report(a, b, rtol, what):
    bad = []
    for k in a:
        if a[k] is None or b[k] is None:
            if not (a[k] is None and b[k] is None):
                bad.append("%s: one side has no value" % k)
            continue
        ref = float(b[k].abs().max()) + 1e-12
        err = max_err(a[k], b[k]) / ref
        if not err < rtol:
            bad.append("%s: rel-max err %.3e (ref max %.3e)" % (k, err, ref))
    assert not bad, what + ":\n  " + "\n  ".join(bad)


@pytest.mark.parametrize("B,H,W,n_dec,drop", [(2, 16, 64, 2, False), (2, 16, 64, 2, True), (3, 4, 20, 2, True), (1, 2, 5, 1, True),
                                              (5, 16, 64, 1, False)])
def test_fused_stack_equals_operator_chain(dev, B, H, W, n_dec, drop):
    ig = _interp(H, W, n_dec).to(dev)
    g = torch.Generator().manual_seed(B + H)
    feat = torch.randn(B, H, W, 64, generator=g)
    tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1)
    qpos = torch.randn(B, H * W, 64, generator=g) * 0.5
    a = _run(ig, feat, tp, qpos, True, drop, dev)
    b = _run(ig, feat, tp, qpos, False, drop, dev)
    unused = [k for k in a if a[k] is None]
    # the reference's grad-less tensors and the query GRU (its output is an input of this test), nothing else
    assert all(("self_attn" in k and "decoder" in k) or ("decoder.layers" in k and ".norm1." in k) or "fc_feature_in" in k
               or "gru_encoding" in k or "init_factor" in k for k in unused), unused
    _report(a, b, 2e-4, "fused vs operator chain (B=%d H=%d W=%d layers=%d dropout=%s)" % (B, H, W, n_dec, drop))
    if drop:
        assert float((a["wts"] == 0).float().mean()) < 0.05 and abs(float(a["wts"].sum()) / (B * H * W) - 1.0) < 0.1


def test_fused_stack_vs_oracle(dev):
    """Dropout off: forward values and all gradients against the CPU restatement of the reference."""
    B, H, W = 2, 16, 64
    ig = _interp(H, W, 2)
    sd = {"infoGen." + k: v.detach().clone() for k, v in ig.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(B, H, W, 64, generator=g)
    tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1)
    # oracle: qpos comes from the query GRU inside tp_interpreter; feed the same embedding to the HIP path
    qpos = O.query_embedding(sd, "infoGen", B, H, W)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "pe.pe" not in k}
    sdl = dict(sd)
    sdl.update(leaves)
    f_c = feat.clone().requires_grad_(True)
    t_c = tp.clone().requires_grad_(True)
    tp_map_o, wts_o = O.tp_interpreter(f_c.permute(0, 3, 1, 2), t_c, sdl, "infoGen", False)
    gg = torch.Generator().manual_seed(5)
    w1 = torch.randn(B, H, W, 64, generator=gg)
    w2 = torch.randn(wts_o.shape, generator=gg)
    ((tp_map_o.permute(0, 2, 3, 1) * w1).sum() + (wts_o * w2).sum() * 3.0).backward()
    got = _run(ig.to(dev), feat, tp, qpos.detach(), True, False, dev)
    ref = {"tp_map": tp_map_o.permute(0, 2, 3, 1), "wts": wts_o, "d.feat": f_c.grad, "d.tp": t_c.grad}
    for k, v in leaves.items():
        n = k[len("infoGen."):]
        if ("g." + n) in got and got["g." + n] is not None and "gru_encoding" not in n and "init_factor" not in n:
            ref["g." + n] = v.grad
    bad = []
    for k, r in ref.items():
        r = r.detach().float()
        scale = float(r.abs().max()) + 1e-12
        e = max_err(got[k], r) / scale
        if not e < 5e-4:
            bad.append("%s: %.3e" % (k, e))
    assert not bad, "\n".join(bad)
    assert max_err(got["tp_map"], ref["tp_map"]) < 2e-5 and max_err(got["wts"], ref["wts"]) < 1e-6
