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


def _layer_inputs(dev, B, L, S, fin, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
    x, qpos, K, V, up = r(B, L, 64), r(B, L, 64), r(B, S, 64), r(B, S, 64), r(B, L, 64)
    lp = (r(192, 64), r(192), r(64, 64), r(64), r(64, 64), r(64), r(64, 64), r(64), r(64) + 1, r(64), r(64) + 1, r(64))
    lnF = (r(64) + 1, r(64)) if fin else None
    return x, qpos, K, V, up, lp, lnF


@pytest.mark.parametrize("B,L,p,fin", [(48, 1024, 0.1, True), (48, 1024, 0.0, False), (5, 256, 0.1, False), (16, 4096, 0.1, True)])
def test_second_generation_layer_against_the_first(dev, B, L, p, fin):
    """csrc/tplayer2.hip against csrc/tplayer.hip on the same inputs, kernel level, at the benchmarked token count (B = 48: three rounds per
    work-group, work-groups that change sample), the large tile and a small case: the forward (exact fp32, another summation order) to
    fp32 round-off; the backward (split-bf16 products, relu decisions taken from the forward's bits) to 1e-4 of each output's scale --
    every token, so a single flipped relu (O(1) on that token) fails."""
    from tatt_amd import ops, functional as Fh
    S = 26
    x, qpos, K, V, up, lp, lnF = _layer_inputs(dev, B, L, S, fin)
    seed = Fh.seed_tensor(dev)
    assert ops.tplayer2_geom(B, L, S)[0] == 1
    pk = ops.tplayer2_prep(lp, K, V)
    o1 = ops.tplayer_fwd(x, qpos, K, V, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, not fin, fin)
    o2 = ops.tplayer2_fwd(x, qpos, pk, lp, lnF, 0.5, int(fin), p, p, p, seed, 10, 1e-5, not fin, fin, S)
    for n, a, b in zip(("xout", "fin", "wavg"), o2[:3], o1):
        assert (a is None) == (b is None), n
        if a is not None:
            assert max_err(a, b) <= 2e-6 * float(b.abs().max()), (n, max_err(a, b))
    hm = o2[3]
    bargs = (0.5, int(fin), p, p, p, seed, 10, 1e-5, None if fin else up, up if fin else None, None, None, True)
    dx1, dq1, kv1, pp1 = ops.tplayer_bwd(x, qpos, K, V, lp, lnF, *bargs)
    dK1, dV1 = ops.tplayer_reduce_kv(kv1, B, L, S)
    dx2, dq2, kv2, fl2, pp2, G2 = ops.tplayer2_bwd(x, qpos, pk, lp, lnF, *bargs, S, hmask=hm)
    dK2, dV2 = ops.tplayer2_reduce_kv(kv2, fl2, B, L, S)
    shp = [(64, 64), (64,), (64, 64), (64,), (64, 64), (64,), (64, 64), (64,), (64,), (64,), (64,), (64,), (64,), (64,)]
    g1 = [torch.zeros(*sh, device=dev) for sh in shp]
    g2 = [torch.zeros(*sh, device=dev) for sh in shp]
    if not fin:
        g1[12] = g1[13] = g2[12] = g2[13] = None
    ops.tplayer_reduce_params(pp1, B, L, g1)
    ops.tplayer_reduce_params_g(pp2, G2, g2)
    names = ["dx", "dqpos", "dK", "dV", "in_w", "in_b", "out_w", "out_b", "w1", "b1", "w2", "b2", "lnA_w", "lnA_b", "lnB_w", "lnB_b", "lnF_w", "lnF_b"]
    bad = []
    for n, a, b in zip(names, [dx2, dq2, dK2, dV2] + g2, [dx1, dq1, dK1, dV1] + g1):
        if a is None:
            continue
        e = max_err(a, b) / (float(b.abs().max()) + 1e-20)
        if not e <= 1e-4:
            bad.append("%s %.2e" % (n, e))
    assert not bad, bad
    # deterministic: a second launch reproduces every word
    dx3, dq3, kv3, fl3, pp3, _ = ops.tplayer2_bwd(x, qpos, pk, lp, lnF, *bargs, S, hmask=hm)
    assert torch.equal(dx2, dx3) and torch.equal(dq2, dq3) and torch.equal(pp2, pp3)
    assert torch.equal(ops.tplayer2_reduce_kv(kv3, fl3, B, L, S)[0], dK2)


def test_second_generation_is_a_training_path_and_fp32_switch_selects_the_first(dev):
    """The second generation runs where a backward follows (it exists to leave the relu bits and the packed operands for it); evaluation
    and tatt_amd.set_arithmetic('fp32') run the first generation.  Checked through what TPStackFn keeps for its backward."""
    import tatt_amd
    import tatt_amd.tsrn as T
    from tatt_amd import functional as Fh, ops
    ig = _interp(16, 64, 2).to(dev)
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(2, 16, 64, 64, generator=g).to(dev).requires_grad_(True)
    tp = torch.softmax(torch.randn(2, 37, 1, 26, generator=g), 1).to(dev)
    qpos = (torch.randn(2, 1024, 64, generator=g) * 0.5).to(dev)
    ig.dropout_on = False
    Fh.begin_training_forward(dev)
    def stack_node(t):                                      # the TPStackFn node behind the reshaped output
        fn = t.grad_fn
        while fn is not None and not hasattr(fn, "packs"):
            fn = fn.next_functions[0][0]
        assert fn is not None
        return fn
    tp_map, _ = T._tp_interpreter(feat, tp, ig, True, qpos=qpos)
    node = stack_node(tp_map)
    assert all(pk is not None for pk in node.packs) and node.hms[0].dtype == torch.int64
    try:
        tatt_amd.set_arithmetic("fp32")
        assert not ops.TPLAYER_BWD2
        tp_map2, _ = T._tp_interpreter(feat, tp, ig, True, qpos=qpos)
        assert all(pk is None for pk in stack_node(tp_map2).packs)
    finally:
        tatt_amd.set_arithmetic("split_bf16")
    assert max_err(tp_map, tp_map2) < 2e-6                  # (both forwards are exact fp32: summation order only)
    with torch.no_grad():
        ev, _ = T._tp_interpreter(feat.detach(), tp, ig, False, qpos=qpos)
    # evaluation runs the first generation's forward (no backward follows: nothing to leave bits for); fp32 round-off of the small GEMMs
    # around the layers is all that separates it from the training-mode forward with dropout off
    assert max_err(ev, tp_map2.detach()) < 5e-6


def test_fused_kv_projection_and_packing_against_the_separate_launches(dev):
    """tatt_tplayer2_kvprep (mem + pos, both layers' key / value projections, the operand packing: one launch) against linear_fwd +
    tatt_tplayer2_prep per layer: weight images bit-identical, mem + pos exact, the fp32 key / value forms to fp32 round-off (another
    summation order), the bf16 hi + lo forms reconstructed to 2^-16."""
    from tatt_amd import ops
    B, S = 5, 26
    g = torch.Generator().manual_seed(11)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.4).to(dev)
    mem = r(B, S, 64)
    lps = [(r(192, 64), r(192), r(64, 64), r(64), r(64, 64), r(64), r(64, 64), r(64), r(64), r(64), r(64), r(64)) for _ in range(2)]

    def bf16_words(t):                                      # int32 words of two bf16 -> (low, high) as float
        t = t.view(torch.int32)
        lo = ((t & 0xFFFF) << 16).view(torch.float32)
        hi = (t & -65536).view(torch.float32)
        return lo, hi

    for pos in (r(S, 64), r(B, S, 64)):
        kin, packs = ops.tplayer2_kvprep(mem, pos, lps)
        assert torch.equal(kin, mem + pos)
        for lp, pk in zip(lps, packs):
            K = ops.linear_fwd(kin.reshape(-1, 64), lp[0][64:128], lp[1][64:128]).reshape(B, S, 64)
            V = ops.linear_fwd(mem.reshape(-1, 64), lp[0][128:], lp[1][128:]).reshape(B, S, 64)
            ref = ops.tplayer2_prep(lp, K, V)
            assert torch.equal(pk[0], ref[0]) and torch.equal(pk[2], ref[2])                      # weight images
            assert max_err(pk[3], ref[3]) <= 2e-6 * float(ref[3].abs().max())                      # fp32 key / value forms
            # bf16 forms: [form][...][hl][lane][words]: hi + lo of the two paths agree to the split's own resolution
            a, b = pk[1].reshape(B, 4, -1), ref[1].reshape(B, 4, -1)
            for form, nw in ((0, 2), (1, 2), (2, 4), (3, 4)):
                va = [x.reshape(B, -1, 2, 64 * nw) for x in bf16_words(a[:, form])]
                vb = [x.reshape(B, -1, 2, 64 * nw) for x in bf16_words(b[:, form])]
                for xa, xb in zip(va, vb):
                    ra, rb = xa[:, :, 0] + xa[:, :, 1], xb[:, :, 0] + xb[:, :, 1]                  # hi plane + lo plane
                    assert max_err(ra, rb) <= 3e-5 * float(rb.abs().max()), form
