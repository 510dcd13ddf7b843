#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (imported read-only from /root/reference)
and validate the CPU oracle against it.  Runs ONLY in the build container; the fixtures it
writes under tests/golden/ are data (inputs + expected outputs), never reference source.

    python tools/gen_golden.py            # validate oracle vs reference, write tests/golden/*.npz

Cases (SURVEY.md §8c "Fixtures to generate"):
  kat            seed-1234 default-init TATT, eval, B=2  (the survey's known-answer vector)
  tatt_eval_b2   randomised weights, eval forward B=2 (BASELINE config 1)
  tsrn_eval_b2   TSRN (no text prior), eval forward B=2
  tatt_train_b4  train-mode BN, every nn.Dropout in eval mode, B=4: sr, loss, grads, post-Adam weights
  tsrn_train_b3  same for TSRN, B=3
  qgru           query-GRU batch-axis quirk at B=1,2,4
  large_tile     32x128 LR, width=256,height=64, STN=False, eval B=1
  tps            TPS grid + sampler with out-of-range control points
  losses         SemanticLoss (value + gradient) and calculate_psnr
  crnn_b3        CRNN text-prior generator (bicubic+luminance input, eval / train logits, parameter gradients), B=3
  tbsrn_b2       TBSRN variant at LR 16x256 (the only size the reference runs): eval forward + train fwd/bwd, B=2
  large_train_b2 32x128 LR (BASELINE configs[4] geometry), STN=False, train fwd/bwd, B=2
  fp64_error_bars  per-tensor distance of the reference's fp32 gradients from the fp64 gradients (tatt_train_b4 case)
  bench_losses   first-step losses of bench.py's own model/batch (dropout off), by the reference
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from _ref_import import import_reference  # noqa: E402
from oracle import tatt_oracle as O  # noqa: E402
from oracle.fixtures import randomize_state_dict, summarize, make_inputs  # noqa: E402
from tests.util import STRUCTURAL_ZERO_GRAD  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def maxdiff(a, b):
    return float((a.detach() - b.detach()).abs().max())


def build_ref(ref, cls, seed=1234, randomize=True, **kw):
    torch.manual_seed(seed)
    m = getattr(ref, cls)(**kw)
    if randomize:
        m.load_state_dict(randomize_state_dict(m.state_dict()))
    return m


def set_dropout_eval(m):
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.MultiheadAttention)):
            mod.eval()      # MHA reads self.training for its attention dropout
    return m


def np_(t):
    return t.detach().cpu().numpy()


def case_eval(ref, name, cls, B, tatt, report, H=16, W=64, **kw):
    m = build_ref(ref, cls, **kw).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, tp, _ = make_inputs(B, H, W)
    with torch.no_grad():
        if tatt:
            y, w = m(x, tp)
        else:
            y, w = m(x), None
        o = O.generator_forward(sd, x, tp if tatt else None, training=False, tatt=tatt,
                                stn=kw.get("STN", False))
    d = maxdiff(y, o["sr"])
    report.append("%-14s oracle-vs-reference max|dsr| = %.3e" % (name, d))
    assert d < 2e-5, (name, d)
    save = dict(x=np_(x), sr=np_(y), block1=np_(m.block["1"][:, :8]), block7=np_(m.block["7"][:, :8]))
    if tatt:
        dw = maxdiff(w, o["pr_weights"])
        assert dw < 1e-5, (name, dw)
        save.update(tp=np_(tp), pr_weights=np_(w))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def case_train(ref, name, cls, B, tatt, report):
    m = build_ref(ref, cls, scale_factor=2, width=128, height=32, STN=True, mask=True,
                  srb_nums=5, hidden_units=32)
    m.train()
    set_dropout_eval(m)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, tp, hr = make_inputs(B)
    loss_mod = ref_image_loss()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    out = m(x, tp) if tatt else m(x)
    sr = out[0] if tatt else out
    loss = loss_mod(sr, hr).mean() * 100
    opt.zero_grad()
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)
    grads_clipped = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in m.named_parameters()}
    opt.step()
    sd1 = m.state_dict()

    o_loss, o_grads, o_sd1, _, o_out, o_total = O.train_step(sd0, x, tp if tatt else None, hr, tatt=tatt, stn=True)
    dl = abs(float(loss) - float(o_loss))
    dsr = maxdiff(sr, o_out["sr"])
    report.append("%-14s loss ref %.6f oracle %.6f  max|dsr| %.3e  gnorm ref %.5f oracle %.5f" %
                  (name, float(loss), float(o_loss), dsr, float(gnorm), float(o_total)))
    # NOTE conditioning: with STN on, fp32 round-off in the control points (~2e-7) is amplified
    # by the bilinear sampler on a white-noise image (|d img / d coord| ~ W = 64 px per unit),
    # so train-mode tensors agree to ~1e-4, not 1e-6 -- between the reference and ANY re-implementation.
    assert dl < 1e-4 * max(1.0, abs(float(loss))) and dsr < 3e-4, (dl, dsr)
    assert abs(float(gnorm) - float(o_total)) < 1e-3 * float(gnorm)
    coef = min(1.0, 0.25 / (float(gnorm) + 1e-6))
    worst = 0.0
    none_keys = []
    for k, g in grads_clipped.items():
        if g is None:
            assert o_grads[k] is None, k
            none_keys.append(k)
            continue
        og = o_grads[k] * coef
        # conv biases that feed a BatchNorm have a mathematically zero gradient (pure round-off):
        # floor the denominator so those compare as absolute noise.
        rel = float((g - og).norm() / (g.norm() + 1e-6 * g.numel() ** 0.5))
        worst = max(worst, rel)
        assert rel < 1e-2, (k, rel)
    # Adam's first step is lr * g/(|g|+eps'): an element whose gradient is round-off noise moves by
    # +-lr with a noise-determined sign, so post-Adam weights are compared by MEAN |diff| per tensor and
    # the keys whose whole gradient is noise (conv biases in front of a BatchNorm) are listed, not compared.
    wworst = 0.0
    noise_keys = [k for k, g in grads_clipped.items()
                  if g is not None and float(g.norm()) < 1e-6 * g.numel() ** 0.5]
    for k in sd1:
        if k in noise_keys:
            continue
        wworst = max(wworst, float((sd1[k].float() - o_sd1[k].float()).abs().mean()))
    report.append("%-14s worst rel grad err %.3e ; worst mean|w1 - w1_oracle| %.3e ; %d params without grad"
                  % (name, worst, wworst, len(none_keys)))
    assert wworst < 2e-4, wworst
    keys = [k for k in grads_clipped if grads_clipped[k] is not None]
    save = dict(x=np_(x), hr=np_(hr), sr=np_(sr), loss=np.float64(float(loss)), gnorm=np.float64(float(gnorm)),
                grad_keys=np.array(keys), none_keys=np.array(none_keys), noise_keys=np.array(noise_keys),
                grad_summary=np.stack([summarize(grads_clipped[k] / coef) for k in keys]),
                w1_keys=np.array(list(sd1.keys())),
                w1_summary=np.stack([summarize(sd1[k].float()) for k in sd1]))
    if tatt:
        save.update(tp=np_(tp), pr_weights=np_(out[1]["pr_weights"]),
                    tp_map=np_(out[1]["trans_feat"][:, :8]))
    # a few complete small gradients for element-wise checks
    for k in ("block1.1.weight", "infoGen.fc_in.weight", "block4.gru1.gru.weight_hh_l0",
              "block8.1.bias", "stn_head.stn_fc2.bias", "block2.bn1.weight"):
        if k in grads_clipped and grads_clipped[k] is not None:
            save["g:" + k] = np_(grads_clipped[k] / coef)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def ref_image_loss():
    """loss/image_loss.py needs PIL/IPython/torchvision at import; stubbed like the model imports."""
    import types
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tv.transforms = types.ModuleType("torchvision.transforms")
        sys.modules["torchvision.transforms"] = tv.transforms
    from loss.image_loss import ImageLoss
    return ImageLoss(gradient=True, loss_weight=[1, 1e-4])


def case_qgru(ref, report):
    m = build_ref(ref, "TSRN_TL_TRANS", scale_factor=2, width=128, height=32, STN=False).eval()
    sd = m.state_dict()
    tr = m.infoGen.transformer
    save = {}
    for B in (1, 2, 4):
        q = m.infoGen.init_factor.weight.unsqueeze(1).repeat(1, B, 1)
        q = q.reshape(16, 64, B, 64).permute(1, 2, 0, 3).reshape(64, B, 16 * 64)
        with torch.no_grad():
            q, _ = tr.gru_encoding(q)
        q = q.reshape(64, B, 16, 64).permute(2, 0, 1, 3).reshape(1024, B, 64).permute(1, 0, 2)
        o = O.query_embedding(sd, "infoGen", B, 16, 64)
        d = maxdiff(q, o)
        report.append("qgru B=%d       max|d| = %.3e" % (B, d))
        assert d < 1e-5
        save["q%d" % B] = np_(q[:, ::37])          # every 37th of the 1024 positions
    np.savez_compressed(os.path.join(OUT, "qgru.npz"), **save)


def case_tps(ref, report):
    m = build_ref(ref, "TSRN_TL_TRANS", scale_factor=2, width=128, height=32, STN=True).eval()
    sd = m.state_dict()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 4, 16, 64, generator=g)
    ctrl = sd["stn_head.stn_fc2.bias"].reshape(1, 20, 2) + 0.08 * torch.randn(3, 20, 2, generator=g)
    with torch.no_grad():
        y, src = m.tps(x, ctrl)
    oy, osrc = O.tps_transform(x, ctrl, sd, "tps")
    d = max(maxdiff(y, oy), maxdiff(src, osrc))
    report.append("tps            max|d| = %.3e  (src range %.3f..%.3f)" % (d, float(src.min()), float(src.max())))
    assert d < 1e-5 and float(src.min()) < 0 and float(src.max()) > 1
    np.savez_compressed(os.path.join(OUT, "tps.npz"), x=np_(x), ctrl=np_(ctrl), y=np_(y), src=np_(src))


def case_kat(ref, report):
    torch.manual_seed(1234)
    m = ref.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5,
                          hidden_units=32).eval()
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 4, 16, 64, generator=g)
    tp = torch.softmax(torch.randn(2, 37, 1, 26, generator=g), 1)
    with torch.no_grad():
        y, w = m(x, tp)
    s = float(y.double().sum())
    report.append("kat            sum(y) = %.6f (survey: 168.209915)" % s)
    assert abs(s - 168.209915) < 1e-4
    sd = m.state_dict()
    np.savez_compressed(os.path.join(OUT, "kat.npz"), x=np_(x), tp=np_(tp), sr=np_(y), pr_weights=np_(w),
                        sd_keys=np.array(list(sd.keys())),
                        sd_summary=np.stack([summarize(v.float()) for v in sd.values()]))
    # TSRN default-init fingerprint too (init-order check for the product module)
    torch.manual_seed(1234)
    t = ref.TSRN(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    sd = t.state_dict()
    np.savez_compressed(os.path.join(OUT, "kat_tsrn.npz"), sd_keys=np.array(list(sd.keys())),
                        sd_summary=np.stack([summarize(v.float()) for v in sd.values()]))


def case_tbsrn(ref, report):
    """TBSRN at LR 16x256 (H*W = 4096, the only size the unmodified reference runs): eval forward + train fwd/bwd, B=2."""
    from model import tbsrn as rt
    kw = dict(scale_factor=2, width=512, height=32, STN=True, mask=True, input_channel=4)
    torch.manual_seed(1234)
    m = rt.TBSRN(**kw)
    m.load_state_dict(randomize_state_dict(m.state_dict()))
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, _, hr = make_inputs(2, 16, 256, seed=2)
    m.eval()
    with torch.no_grad():
        y = m(x)
        o = O.tbsrn_forward(sd0, x, training=False)
    d = maxdiff(y, o["sr"])
    report.append("tbsrn_eval_b2  oracle-vs-reference max|dsr| = %.3e" % d)
    assert d < 2e-5, d
    # train mode: the reference's STN head only accepts 16x64 inputs (stn_fc1 takes 512 features) while its FeatureEnhancer
    # only accepts H*W == 4096, so the unmodified reference TBSRN can only TRAIN with the STN bypassed.  m.stn = False does
    # that without changing the parameter set.
    m.train()
    m.stn = False
    set_dropout_eval(m)
    loss_mod = ref_image_loss()
    sr = m(x)
    loss = loss_mod(sr, hr).mean() * 100
    m.zero_grad()
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in m.named_parameters()}
    o_loss, o_grads, _, _, o_out, o_total = O.train_step(sd0, x, None, hr, stn=False, tbsrn=True)
    dsr = maxdiff(sr, o_out["sr"])
    worst = 0.0
    none_keys = []
    scale = max(float(g.abs().max()) for g in grads.values() if g is not None)
    for k, g in grads.items():
        if g is None:
            assert o_grads[k] is None, k
            none_keys.append(k)
            continue
        # conv biases in front of a BatchNorm: mathematically zero gradient, pure round-off -> floor the denominator
        rel = float((g - o_grads[k]).norm() / (g.norm() + 1e-6 * scale * g.numel() ** 0.5))
        worst = max(worst, rel)
        assert rel < 2e-2, (k, rel)
    report.append("tbsrn_train_b2 loss ref %.6f oracle %.6f  max|dsr| %.3e  worst rel grad err %.3e ; %d params without grad"
                  % (float(loss), float(o_loss), dsr, worst, len(none_keys)))
    assert abs(float(loss) - float(o_loss)) < 1e-4 * float(loss) and dsr < 3e-4
    keys = [k for k in grads if grads[k] is not None]
    np.savez_compressed(os.path.join(OUT, "tbsrn_b2.npz"), x=np_(x), hr=np_(hr), sr_eval=np_(y), sr_train=np_(sr),
                        loss=np.float64(float(loss)), grad_keys=np.array(keys), none_keys=np.array(none_keys),
                        grad_summary=np.stack([summarize(grads[k]) for k in keys]),
                        sd_keys=np.array(list(m.state_dict().keys())))
    torch.manual_seed(1234)
    fresh = rt.TBSRN(**kw).state_dict()
    np.savez_compressed(os.path.join(OUT, "kat_tbsrn.npz"), sd_keys=np.array(list(fresh.keys())),
                        sd_summary=np.stack([summarize(v.float()) for v in fresh.values()]))


def case_crnn(ref, report):
    """CRNN text-prior generator: parse_crnn_data + CRNN(32,1,37,256) eval / train forward + backward, B = 3."""
    from model.crnn import crnn as rc
    from oracle import crnn_oracle as C
    torch.manual_seed(1234)
    fresh = rc.CRNN(32, 1, 37, 256).state_dict()
    torch.manual_seed(1234)
    m = rc.CRNN(32, 1, 37, 256)
    m.load_state_dict(randomize_state_dict(m.state_dict()))
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    img = torch.rand(3, 4, 16, 64, generator=g)
    # reference interfaces/base.py:797-815 (parse_crnn_data), restated inline because TextBase needs the whole config stack
    r = torch.nn.functional.interpolate(img[:, :3], (32, 100), mode="bicubic")
    x_ref = 0.299 * r[:, 0:1] + 0.587 * r[:, 1:2] + 0.114 * r[:, 2:3]
    x = C.parse_crnn_data(img)
    d_in = maxdiff(x, x_ref)
    m.eval()
    with torch.no_grad():
        y_eval = m(x_ref)
        o_eval = C.crnn_forward(sd0, x)
    m.train()
    wts = torch.randn(26, 3, 37, generator=g)
    y = m(x_ref)
    prior = torch.softmax(y, -1)
    (prior * wts).sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    sd_req = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd0.items()}
    stats = {}
    o = C.crnn_forward(sd_req, x, training=True, new_stats=stats)
    (torch.softmax(o, -1) * wts).sum().backward()
    worst = 0.0
    scale = max(float(gr.abs().max()) for gr in grads.values())
    for k, gr in grads.items():
        rel = float((gr - sd_req[k].grad).norm() / (gr.norm() + 1e-6 * scale * gr.numel() ** 0.5))
        # conditioning: from conv6 on the two agree to 1e-6; below the train-mode BatchNorm4 (B = 3) the fp32 gradient of the
        # reference ITSELF is 1.6e-2 from its own fp64 gradient at conv0 (ReLU / max-pool selections flip on round-off).
        if k in ("cnn.conv2.bias", "cnn.conv4.bias", "cnn.conv6.bias"):      # bias in front of a BatchNorm: structurally zero gradient
            assert float(gr.abs().max()) < 1e-5 * scale and float(sd_req[k].grad.abs().max()) < 1e-5 * scale, k
            continue
        deep = k.startswith("rnn.") or any(k.startswith("cnn.%s" % n) for n in ("conv6", "batchnorm6"))
        worst = max(worst, rel if deep else 0.0)
        assert rel < (1e-4 if deep else 5e-2), (k, rel)
    report.append("crnn_b3        input max|d| %.2e  eval logits max|d| %.2e  train logits max|d| %.2e  worst rel grad err (conv6..rnn) %.2e"
                  % (d_in, maxdiff(y_eval, o_eval), maxdiff(y, o), worst))
    assert d_in < 2e-6 and maxdiff(y_eval, o_eval) < 1e-6 and maxdiff(y, o) < 1e-5
    sd1 = m.state_dict()
    keys = list(grads.keys())
    np.savez_compressed(os.path.join(OUT, "crnn_b3.npz"), img=np_(img), x=np_(x_ref), wts=np_(wts), logits_eval=np_(y_eval),
                        logits_train=np_(y), prior=np_(prior.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)),
                        grad_keys=np.array(keys), grad_summary=np.stack([summarize(grads[k]) for k in keys]),
                        **{"g:" + k: np_(grads[k]) for k in ("cnn.conv0.weight", "cnn.batchnorm6.weight", "rnn.1.embedding.bias",
                                                            "rnn.0.rnn.bias_hh_l0_reverse")},
                        bn_mean=np_(sd1["cnn.batchnorm4.running_mean"]), bn_var=np_(sd1["cnn.batchnorm4.running_var"]),
                        sd_keys=np.array(list(fresh.keys())),
                        sd_summary=np.stack([summarize(v.float()) for v in fresh.values()]))


def case_losses(ref, report):
    """SemanticLoss (loss/semantic_loss.py) on random student / teacher priors with its gradient, calculate_psnr (utils/ssim_psnr.py)."""
    ref_image_loss()                                   # installs the PIL / torchvision stubs the loss modules import
    from loss.semantic_loss import SemanticLoss
    g = torch.Generator().manual_seed(13)
    pred = torch.softmax(torch.randn(4, 26, 37, generator=g), -1).requires_grad_(True)
    gt = torch.softmax(2 * torch.randn(4, 26, 37, generator=g), -1)
    loss = SemanticLoss()(pred, gt)
    loss.backward()
    o_pred = pred.detach().clone().requires_grad_(True)
    o = O.semantic_loss(o_pred, gt)
    o.backward()
    a, b = torch.rand(3, 4, 32, 128, generator=g), torch.rand(3, 4, 32, 128, generator=g)
    # reference utils/ssim_psnr.py:9-15, restated inline (the module imports cv2 at load)
    mse = ((a[:, :3] * 255 - b[:, :3] * 255) ** 2).mean()
    psnr = 20 * torch.log10(255.0 / torch.sqrt(mse))
    # 8f-2 (oracle side): SSIM / TRI_SSIM (utils/ssim_psnr.py needs only IPython, already stubbed) and torch_distortion
    from utils import ssim_psnr as sp
    from model import torch_distortion as ref_distort
    c = torch.rand(3, 4, 32, 128, generator=g)
    arcs = (torch.rand(3, generator=g) - 0.5) * 0.2
    offs = torch.rand(3, generator=g)
    s_ref, t_ref = sp.SSIM()(a, b), sp.TRI_SSIM()(a, b, c)
    s_ref_b = sp.SSIM(size_average=False)(a, b)
    d_ref = ref_distort(a, arcs, offs)
    ds = max(abs(float(s_ref) - float(O.ssim(a, b))), abs(float(t_ref) - float(O.tri_ssim(a, b, c))),
             maxdiff(s_ref_b, O.ssim(a, b, size_average=False)))
    dd = maxdiff(d_ref, O.torch_distortion(a, arcs, offs))
    report.append("ssim/rotate    |ssim, tri_ssim diff| %.2e ; torch_distortion max|d| %.2e" % (ds, dd))
    assert ds < 1e-6 and dd < 1e-5
    extra = dict(c=np_(c), arcs=np_(arcs), offs=np_(offs), ssim=np.float64(float(s_ref)), tri_ssim=np.float64(float(t_ref)),
                 ssim_per_sample=np_(s_ref_b), distorted=np_(d_ref[:, :, ::4, ::4]))
    report.append("losses         semantic loss ref %.7f oracle %.7f  max|dgrad| %.2e ; psnr ref %.5f oracle %.5f"
                  % (float(loss), float(o), maxdiff(pred.grad, o_pred.grad), float(psnr), float(O.calculate_psnr(a, b))))
    assert abs(float(loss) - float(o)) < 1e-6 and maxdiff(pred.grad, o_pred.grad) < 1e-8
    np.savez_compressed(os.path.join(OUT, "losses.npz"), pred=np_(pred), gt=np_(gt), sem=np.float64(float(loss)),
                        dpred=np_(pred.grad), a=np_(a), b=np_(b), psnr=np.float64(float(psnr)), **extra)


def case_train_large(ref, report):
    """BASELINE.json configs[4] geometry: TSRN_TL_TRANS(2, 256, 64, STN=False) on LR 32x128, train-mode BN, every nn.Dropout in eval
    mode, B = 2: SR, loss, gradient fingerprints (+ a few complete small gradients), BatchNorm running statistics."""
    kw = dict(scale_factor=2, width=256, height=64, STN=False, mask=True, srb_nums=5, hidden_units=32)
    m = build_ref(ref, "TSRN_TL_TRANS", **kw)
    m.train()
    set_dropout_eval(m)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, tp, hr = make_inputs(2, 32, 128, seed=11)
    loss_mod = ref_image_loss()
    sr, mid = m(x, tp)
    loss = loss_mod(sr, hr).mean() * 100
    m.zero_grad()
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in m.named_parameters()}
    o_loss, o_grads, o_sd1, _, o_out, o_total = O.train_step(sd0, x, tp, hr, tatt=True, stn=False)
    dsr = maxdiff(sr, o_out["sr"])
    worst, none_keys = 0.0, []
    scale = max(float(g.abs().max()) for g in grads.values() if g is not None)
    for k, g in grads.items():
        if g is None:
            assert o_grads[k] is None, k
            none_keys.append(k)
            continue
        if STRUCTURAL_ZERO_GRAD.match(k):          # conv bias in front of a BatchNorm: mathematically zero gradient, pure round-off
            assert float(g.abs().max()) < 1e-4 * scale and float(o_grads[k].abs().max()) < 1e-4 * scale, k
            continue
        rel = float((g - o_grads[k]).norm() / (g.norm() + 1e-6 * scale * g.numel() ** 0.5))
        worst = max(worst, rel)
        assert rel < 5e-3, (k, rel)
    report.append("large_train_b2 loss ref %.6f oracle %.6f  max|dsr| %.3e  worst rel grad err %.3e ; %d params without grad"
                  % (float(loss), float(o_loss), dsr, worst, len(none_keys)))
    assert abs(float(loss) - float(o_loss)) < 1e-5 * float(loss) and dsr < 2e-5
    keys = [k for k in grads if grads[k] is not None]
    sd1 = m.state_dict()
    save = dict(x=np_(x), tp=np_(tp), hr=np_(hr), sr=np_(sr), loss=np.float64(float(loss)),
                pr_weights=np_(mid["pr_weights"][:, ::16]), tp_map=np_(mid["trans_feat"][:, :4]),
                grad_keys=np.array(keys), none_keys=np.array(none_keys),
                grad_summary=np.stack([summarize(grads[k]) for k in keys]),
                bn_mean=np_(sd1["block4.bn1.running_mean"]), bn_var=np_(sd1["block4.bn1.running_var"]))
    for k in ("block1.1.weight", "infoGen.fc_in.weight", "block4.gru1.gru.weight_hh_l0", "block4.gru2.gru.weight_ih_l0_reverse",
              "block8.1.bias", "block2.bn1.weight", "infoGen.transformer.decoder.layers.1.norm3.weight",
              "infoGen.transformer.gru_encoding.bias_hh_l0", "infoGen.transformer.decoder.layers.0.multihead_attn.in_proj_bias"):
        save["g:" + k] = np_(grads[k])
    np.savez_compressed(os.path.join(OUT, "large_train_b2.npz"), **save)


def case_fp64_error_bars(ref, report, B=4, fname="fp64_error_bars.npz", seed=None):
    """How far is the REFERENCE's own fp32 gradient from the fp64 gradient of the same graph?  The tatt_train_b4 case again (same seed,
    weights and inputs): reference fp32 backward vs the oracle evaluated in fp64.  Stores, per parameter tensor,
    ||g_ref32 - g_64|| / ||g_64|| -- the yardstick tests/test_model_gpu.py::test_gradients_vs_fp64 holds the HIP gradients to
    (the fp64 gradients themselves are recomputed by the oracle on the test host: 60 MB would be too much to commit).
    B = 48 (`fp64_error_bars_b48.npz`, inputs make_inputs(48, seed=48)): the same yardstick at the benchmarked batch, STN on."""
    m = build_ref(ref, "TSRN_TL_TRANS", scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    m.train()
    set_dropout_eval(m)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, tp, hr = make_inputs(B) if seed is None else make_inputs(B, seed=seed)
    loss = ref_image_loss()(m(x, tp)[0], hr).mean() * 100
    m.zero_grad()
    loss.backward()
    g32 = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    l64, g64, _, _, o64, _ = O.train_step_fp64(sd0, x, tp, hr, tatt=True, stn=True)
    _, o32, _, _, _, _ = O.train_step(sd0, x, tp, hr, tatt=True, stn=True)
    scale = max(float(g.abs().max()) for g in g64.values() if g is not None)
    keys, ref_err, ora_err = [], [], []
    for k, g in g32.items():
        d = g64[k]
        den = float(d.norm()) + 1e-7 * scale * d.numel() ** 0.5
        keys.append(k)
        ref_err.append(float((g.double() - d).norm()) / den)
        ora_err.append(float((o32[k].double() - d).norm()) / den)
    i = int(np.argmax(ref_err))
    report.append("fp64 yardstick B=%d loss fp64 %.7f (ref fp32 %.6f); worst ||g_ref32 - g_64||/||g_64|| = %.3e (%s); oracle fp32: %.3e"
                  % (B, float(l64), float(loss), ref_err[i], keys[i], max(ora_err)))
    np.savez_compressed(os.path.join(OUT, fname), keys=np.array(keys), ref32_err=np.array(ref_err),
                        oracle32_err=np.array(ora_err), loss64=np.float64(float(l64)), scale=np.float64(scale))


def case_bench_losses(ref, report):
    """Known-answer losses for bench.py: the benchmark's own model (seed-1234 default init) and batch (data seed 0), FIRST training-step
    loss with every nn.Dropout in eval mode, computed by the reference itself.  bench.py replays this forward on the GPU before its
    timed region and refuses to print a number if the loss is off."""
    import json
    out = {}
    loss_mod = ref_image_loss()
    for name, kw, B, H, W in (("tatt_b48_16x64", dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32), 48, 16, 64),
                              ("tatt_b16_32x128", dict(scale_factor=2, width=256, height=64, STN=False, mask=True, srb_nums=5, hidden_units=32), 16, 32, 128)):
        torch.manual_seed(1234)
        m = ref.TSRN_TL_TRANS(**kw).train()
        set_dropout_eval(m)
        g = torch.Generator().manual_seed(0)                      # bench.py make_batch(rank 0)
        x = torch.rand(B, 4, H, W, generator=g)
        x[:, 3] = (x[:, 3] > 0.5).float()
        hr = torch.rand(B, 4, 2 * H, 2 * W, generator=g)
        tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1)
        with torch.no_grad():
            sr, _ = m(x, tp)
            loss = float(loss_mod(sr, hr).mean() * 100)
        out[name] = loss
        report.append("bench loss     %-16s first-step loss (dropout off) = %.6f" % (name, loss))
    with open(os.path.join(OUT, "bench_losses.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    report = ["golden vectors generated from /root/reference (torch %s, CPU fp32)" % torch.__version__]
    if "--only-fp64-b48" in sys.argv:            # one case alone (appends its line to REPORT.txt)
        case_fp64_error_bars(ref, report, B=48, fname="fp64_error_bars_b48.npz", seed=48)
        with open(os.path.join(OUT, "REPORT.txt"), "a") as f:
            f.write(report[-1] + "\n")
        print(report[-1])
        return
    case_kat(ref, report)
    std = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    case_eval(ref, "tatt_eval_b2", "TSRN_TL_TRANS", 2, True, report, **std)
    case_eval(ref, "tsrn_eval_b2", "TSRN", 2, False, report, **std)
    case_eval(ref, "large_tile", "TSRN_TL_TRANS", 1, True, report, H=32, W=128,
              scale_factor=2, width=256, height=64, STN=False, mask=True, srb_nums=5, hidden_units=32)
    case_train(ref, "tatt_train_b4", "TSRN_TL_TRANS", 4, True, report)
    case_train(ref, "tsrn_train_b3", "TSRN", 3, False, report)
    case_qgru(ref, report)
    case_tps(ref, report)
    case_tbsrn(ref, report)
    case_crnn(ref, report)
    case_losses(ref, report)
    case_train_large(ref, report)
    case_fp64_error_bars(ref, report)
    case_fp64_error_bars(ref, report, B=48, fname="fp64_error_bars_b48.npz", seed=48)
    case_bench_losses(ref, report)
    with open(os.path.join(OUT, "REPORT.txt"), "w") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
