"""CRNN text-prior generator: oracle vs the reference-generated vectors (CPU) and the HIP path vs both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import crnn_oracle as C
from oracle.fixtures import randomize_state_dict, summarize
from tests.util import max_err, rel_err, check_close

STRUCT_ZERO = ("cnn.conv2.bias", "cnn.conv4.bias", "cnn.conv6.bias")       # biases in front of a BatchNorm


def _is_deep(k):
    return k.startswith("rnn.") or k.startswith("cnn.conv6") or k.startswith("cnn.batchnorm6")


def _sd(randomize=True):
    import tatt_amd
    torch.manual_seed(1234)
    sd = tatt_amd.CRNN(32, 1, 37, 256).state_dict()
    return randomize_state_dict(sd) if randomize else sd


def test_crnn_state_dict_matches_reference():
    z = np.load("tests/golden/crnn_b3.npz")
    sd = _sd(randomize=False)
    assert list(sd.keys()) == z["sd_keys"].tolist() and len(sd) == 49
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == 8331301
    for k, ref in zip(sd, z["sd_summary"]):
        assert np.abs(summarize(sd[k].float()) - ref).max() < 1e-6, k


def test_crnn_oracle_against_reference_vectors():
    z = np.load("tests/golden/crnn_b3.npz")
    sd = _sd()
    img = torch.from_numpy(z["img"])
    x = C.parse_crnn_data(img)
    assert max_err(x, torch.from_numpy(z["x"])) < 2e-6
    with torch.no_grad():
        assert max_err(C.crnn_forward(sd, x), torch.from_numpy(z["logits_eval"])) < 1e-6
    req = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    stats = {}
    y = C.crnn_forward(req, x, training=True, new_stats=stats)
    assert max_err(y, torch.from_numpy(z["logits_train"])) < 1e-5
    assert max_err(C.text_prior(y), torch.from_numpy(z["prior"])) < 1e-6
    (torch.softmax(y, -1) * torch.from_numpy(z["wts"])).sum().backward()
    for k, ref in zip(z["grad_keys"].tolist(), z["grad_summary"]):
        if k in STRUCT_ZERO:
            continue
        got = summarize(req[k].grad)
        lim = 1e-4 if _is_deep(k) else 5e-2            # conditioning below the train-mode BatchNorms: tests/golden/REPORT.txt
        assert abs(got[0] - ref[0]) < lim * ref[0], (k, got[0], ref[0])
    assert max_err(stats["cnn.batchnorm4.running_mean"], torch.from_numpy(z["bn_mean"])) < 1e-6
    assert max_err(stats["cnn.batchnorm4.running_var"], torch.from_numpy(z["bn_var"])) < 1e-6


def test_crnn_has_no_cpu_fallback():
    import tatt_amd
    from tatt_amd.crnn import parse_crnn_data
    with pytest.raises(RuntimeError, match="GPU"):
        tatt_amd.CRNN(32, 1, 37, 256).eval()(torch.rand(1, 1, 32, 100))
    with pytest.raises(RuntimeError, match="GPU"):
        parse_crnn_data(torch.rand(1, 3, 16, 64))


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_crnn_input_and_eval_logits(dev):
    import tatt_amd
    from tatt_amd.crnn import parse_crnn_data
    z = np.load("tests/golden/crnn_b3.npz")
    m = tatt_amd.CRNN(32, 1, 37, 256)
    m.load_state_dict(_sd())
    m = m.to(dev).eval()
    x = parse_crnn_data(torch.from_numpy(z["img"]).to(dev))
    assert tuple(x.shape) == (3, 1, 32, 100)
    assert max_err(x, torch.from_numpy(z["x"])) < 5e-6
    with torch.no_grad():
        y = m(x)
    assert tuple(y.shape) == (26, 3, 37)
    assert max_err(y, torch.from_numpy(z["logits_eval"])) < 2e-6


@pytest.mark.gpu
def test_crnn_train_forward_backward(dev):
    import tatt_amd
    from tatt_amd.crnn import text_prior
    z = np.load("tests/golden/crnn_b3.npz")
    m = tatt_amd.CRNN(32, 1, 37, 256)
    sd0 = _sd()
    m.load_state_dict(sd0)
    m = m.to(dev).train()
    x = torch.from_numpy(z["x"]).to(dev)
    y = m(x)
    assert max_err(y, torch.from_numpy(z["logits_train"])) < 1e-5
    prior = text_prior(y)
    assert tuple(prior.shape) == (3, 37, 1, 26)
    assert max_err(prior, torch.from_numpy(z["prior"])) < 1e-6
    w = torch.from_numpy(z["wts"]).to(dev)
    (prior * w.permute(1, 2, 0).unsqueeze(2)).sum().backward()
    params = dict(m.named_parameters())
    scale = max(float(r[0]) for r in z["grad_summary"])
    for k, ref in zip(z["grad_keys"].tolist(), z["grad_summary"]):
        g = params[k].grad
        assert g is not None, k
        if k in STRUCT_ZERO:
            assert float(g.abs().max()) < 1e-4 * scale, k
            continue
        got = summarize(g.cpu())
        lim = 2e-4 if _is_deep(k) else 5e-2
        assert abs(got[0] - ref[0]) < lim * ref[0], (k, got[0], ref[0])
    for key in z.files:
        if key.startswith("g:"):
            k = key[2:]
            assert rel_err(params[k].grad.cpu(), torch.from_numpy(z[key])) < (2e-4 if _is_deep(k) else 5e-2), k
    sd1 = m.state_dict()
    assert max_err(sd1["cnn.batchnorm4.running_mean"], torch.from_numpy(z["bn_mean"])) < 1e-5
    assert max_err(sd1["cnn.batchnorm4.running_var"], torch.from_numpy(z["bn_var"])) < 1e-5
    assert int(sd1["cnn.batchnorm4.num_batches_tracked"]) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("T,Bt,I,H", [(5, 3, 64, 256), (26, 48, 512, 256), (7, 17, 256, 256)])
def test_bilstm_kernels(dev, T, Bt, I, H):
    from tatt_amd import functional as Fh
    from tests.util import compare_fn
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, Bt, I, generator=g)
    ws = []
    for _ in range(2):
        ws += [torch.randn(4 * H, I, generator=g) / I ** 0.5, torch.randn(4 * H, H, generator=g) / H ** 0.5,
               torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1]

    def ref(x, *w):
        return torch.cat([C.lstm_direction(x, *w[:4], False), C.lstm_direction(x, *w[4:], True)], -1)
    compare_fn("bilstm", lambda x, *w: Fh.BiLSTMFn.apply(x, *w), ref, [x] + ws, dev, rtol=5e-4, atol=5e-5, grtol=2e-3, gatol=2e-4)


@pytest.mark.gpu
def test_general_maxpool_and_valid_conv(dev):
    from tatt_amd import functional as Fh
    from tests.util import compare_fn
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 4, 27, 16, generator=g)
    for k, s, p in (((2, 2), (2, 1), (0, 1)), ((2, 2), (2, 2), (0, 0)), ((3, 2), (1, 2), (1, 0))):
        compare_fn("maxpool_general", lambda x: Fh.max_pool(x, k[0], k[1], s[0], s[1], p[0], p[1]),
                   lambda x: torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1), [x], dev)
    # 3x3 convolution at a CRNN geometry (width not a multiple of 64 -> implicit-GEMM kernel, aligned im2col fast paths
    # for forward, data gradient and weight gradient; 2*8*25 = 400 pixels: ragged last row tile)
    x = torch.randn(2, 8, 25, 64, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(128, generator=g)
    compare_fn("conv3_crnn_geometry", lambda x, w, b: Fh.conv2d(x, w, b, 1),
               lambda x, w, b: torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)).permute(0, 2, 3, 1),
               [x, w, b], dev, grtol=1e-3, gatol=1e-4)
    x = torch.randn(3, 2, 27, 64, generator=g)
    w = torch.randn(96, 64, 2, 2, generator=g) * 0.1
    b = torch.randn(96, generator=g)
    compare_fn("conv2x2_valid", lambda x, w, b: Fh.Conv2x2ValidFn.apply(x, w, b),
               lambda x, w, b: torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b).permute(0, 2, 3, 1), [x, w, b], dev,
               grtol=1e-3, gatol=1e-4)


@pytest.mark.gpu
def test_sr_loss_trains_the_prior_generator(dev):
    """TextPriorSR(detach_prior=False) = the reference's `tsrn_tl` composition (interfaces/super_resolution.py:729-768): lr image -> CRNN
    student -> softmax prior -> TSRN_TL_TRANS -> ImageLoss; the SR output and the gradients that reach the recogniser through the prior
    against the oracle composition (dropout off, STN off for conditioning)."""
    import tatt_amd
    from oracle import tatt_oracle as O
    from oracle.fixtures import make_inputs
    from tatt_amd.train import TextPriorSR, image_loss
    kw = dict(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32)
    torch.manual_seed(1234)
    sr_m = tatt_amd.TSRN_TL_TRANS(**kw)
    sr_m.load_state_dict(randomize_state_dict(sr_m.state_dict()))
    tpg = tatt_amd.CRNN(32, 1, 37, 256)
    tpg.load_state_dict(_sd())
    teacher = tatt_amd.CRNN(32, 1, 37, 256).eval()
    sd_teacher = randomize_state_dict(teacher.state_dict(), seed=5)
    teacher.load_state_dict(sd_teacher)
    sd_sr = {k: v.detach().clone() for k, v in sr_m.state_dict().items()}
    sd_tpg = {k: v.detach().clone() for k, v in tpg.state_dict().items()}
    m = TextPriorSR(sr_m, tpg, teacher=teacher, detach_prior=False).to(dev).train()
    assert not teacher.training and len([k for k, _ in m.named_parameters() if "teacher" in k]) == 0
    sr_m.infoGen.dropout_on = False
    x, _, hr = make_inputs(3, seed=11)
    sr, mid = m(x.to(dev))
    loss = image_loss(sr, hr.to(dev)).mean() * 100 + m.extra_loss(hr.to(dev))
    loss.backward()
    # oracle composition
    req = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd_tpg.items()}
    prior = C.text_prior(C.crnn_forward(req, C.parse_crnn_data(x), training=True))
    full = dict(sd_sr)
    out = O.generator_forward(full, x, prior, training=True, tatt=True, stn=False, drop_on=False)
    with torch.no_grad():
        gt = torch.softmax(C.crnn_forward(sd_teacher, C.parse_crnn_data(hr), training=False), -1)
    o_loss = O.image_loss(out["sr"], hr).mean() * 100 + O.semantic_loss(prior.permute(3, 0, 1, 2).squeeze(3), gt) * 100
    o_loss.backward()
    assert max_err(sr, out["sr"].detach()) < 5e-5
    assert abs(float(loss.detach()) - float(o_loss)) < 1e-4 * abs(float(o_loss))
    params = dict(tpg.named_parameters())
    for k in ("rnn.1.embedding.weight", "rnn.1.rnn.weight_hh_l0", "rnn.0.embedding.bias", "rnn.0.rnn.weight_ih_l0_reverse",
              "cnn.conv6.weight", "cnn.batchnorm6.weight"):
        assert params[k].grad is not None, k
        assert rel_err(params[k].grad.cpu(), req[k].grad) < 5e-3, (k, rel_err(params[k].grad.cpu(), req[k].grad))


@pytest.mark.gpu
def test_text_prior_sr_with_teacher_through_the_trainer(dev):
    """The student recogniser receives gradient from the distillation loss AND from the SR generator's text encoder; in the staged
    backward both arrive at stage "tpg" (TextPriorSR.forward cuts there).  One step through the Trainer, eager and as a hipGraph:
    first loss = image loss + distillation term of a plain forward; gradient of a recogniser weight = the single-pass one."""
    import tatt_amd
    from oracle.fixtures import make_inputs
    from tatt_amd.train import TextPriorSR, Trainer, image_loss_mean
    kw = dict(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32)

    def build():
        torch.manual_seed(1234)
        sr_m = tatt_amd.TSRN_TL_TRANS(**kw)
        sr_m.load_state_dict(randomize_state_dict(sr_m.state_dict()))
        tpg = tatt_amd.CRNN(32, 1, 37, 256)
        tpg.load_state_dict(_sd())
        teacher = tatt_amd.CRNN(32, 1, 37, 256)
        teacher.load_state_dict(randomize_state_dict(teacher.state_dict(), seed=5))
        m = TextPriorSR(sr_m, tpg, teacher=teacher, detach_prior=False).to(dev).train()
        sr_m.infoGen.dropout_on = False
        return m
    x, _, hr = make_inputs(3, seed=11)
    x, hr = x.to(dev), hr.to(dev)
    m = build()
    sr, _ = m(x)
    loss = image_loss_mean(sr, hr, scale=100.0) + m.extra_loss(hr)
    loss.backward()                                              # single pass, no staging
    want, g_ref = float(loss), m.tpg.rnn[1].embedding.weight.grad.clone()
    m = build()
    tr = Trainer(m, use_graph=False)
    assert tr.stages[-1] == "tpg" and tr.groups[1][2] == 0.0
    got = float(tr.step(x, None, hr))
    assert abs(got - want) < 1e-6 * abs(want), (got, want)
    assert rel_err(m.tpg.rnn[1].embedding.weight.grad, g_ref) < 1e-5
    # the teacher's pass ran ahead on its own stream, beside the generator's forward (TextPriorSR.begin_teacher) -- same loss without
    assert bool(m._teacher_streams) == tr.two_lanes
    import tatt_amd.train as T
    T.TEACHER_AHEAD = False
    try:
        m2 = build()
        got2 = float(Trainer(m2, use_graph=False).step(x, None, hr))
        assert not m2._teacher_streams and got2 == got
    finally:
        T.TEACHER_AHEAD = True
    # the hook that also forks the student's pass (its backward then runs on that branch's stream): same loss, same gradients
    T.STUDENT_FORK = True
    try:
        m3 = build()
        got3 = float(Trainer(m3, use_graph=False).step(x, None, hr))
        assert got3 == got and rel_err(m3.tpg.rnn[1].embedding.weight.grad, m.tpg.rnn[1].embedding.weight.grad) == 0.0
    finally:
        T.STUDENT_FORK = False
    m = build()
    tr = Trainer(m, use_graph=True, warmup_eager=2)
    ls = [float(tr.step(x, None, hr)) for _ in range(5)]
    assert abs(ls[0] - want) < 1e-6 * abs(want) and all(l == l for l in ls) and ls[-1] < ls[0]


@pytest.mark.gpu
def test_tssim_recipe_with_text_prior_generator_and_plain_tsrn(dev):
    """The shipped configuration (train_TATT.sh: --arch tatt --use_distill --tssim_loss --rotate_train=5), reference
    interfaces/super_resolution.py:770-914: the student recogniser reads the ROTATED LR image ONCE per step; the SR generator gets that
    prior DETACHED in both of its forwards (:873 and :911 pass `label_vecs_final.detach()`), so the student learns from the
    distillation term alone (student prior on x_rot vs teacher prior on hr_rot, x100, :879).  Trainer step (staged) == that composition
    written out operator by operator with a single-pass backward; the recipe also drives generators that take no prior (TSRN)."""
    import tatt_amd
    from oracle.fixtures import make_inputs
    from tatt_amd import functional as Fh
    from tatt_amd.losses import TRI_SSIM
    from tatt_amd.train import TextPriorSR, Trainer, TssimRecipe, image_loss_mean, semantic_loss
    kw = dict(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32)

    def build():
        torch.manual_seed(1234)
        sr_m = tatt_amd.TSRN_TL_TRANS(**kw)
        sr_m.load_state_dict(randomize_state_dict(sr_m.state_dict()))
        tpg = tatt_amd.CRNN(32, 1, 37, 256)
        tpg.load_state_dict(_sd())
        teacher = tatt_amd.CRNN(32, 1, 37, 256)
        teacher.load_state_dict(randomize_state_dict(teacher.state_dict(), seed=5))
        m = TextPriorSR(sr_m, tpg, teacher=teacher).to(dev).train()
        assert m.detach_prior                                   # the default is the reference's `tatt` composition
        sr_m.infoGen.dropout_on = False
        return m
    x, _, hr = make_inputs(3, seed=11)
    x, hr = x.to(dev), hr.to(dev)
    # written out: same angles as the recipe will draw
    rec = TssimRecipe(5.0, seed=4)
    rec.new_step(x)
    rot = Fh.AffineSampleFn.apply
    m = build()
    with torch.no_grad():
        x_rot, hr_rot = rot(x, rec.theta_pos), rot(hr, rec.theta_pos)
        x_ret = rot(x_rot, rec.theta_neg)
        gt = m._probs(m._teacher, hr_rot)
    label_vecs = m._probs(m.tpg, x_rot)                                           # the ONE student pass of the step (T, B, 37)
    label_vecs_final = label_vecs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
    sr = m.sr(x_rot, label_vecs_final.detach())[0]                                # :873
    sr_ret = m.sr(x_ret, label_vecs_final.detach())[0]                            # :911
    dist = semantic_loss(label_vecs, gt) * 100.0                                  # :879
    loss = image_loss_mean(sr, hr_rot, scale=100.0) + (1.0 - TRI_SSIM()(rot(sr_ret, rec.theta_pos), sr, hr_rot)) * 10.0 + dist
    loss.backward()
    want, g_ref = float(loss), m.tpg.rnn[1].embedding.weight.grad.clone()
    g_sr_ref = m.sr.block2.conv1.weight.grad.clone()
    assert float(dist) > 0.0
    # the student's gradient is the distillation term's alone
    m2 = build()
    (semantic_loss(m2._probs(m2.tpg, x_rot), gt) * 100.0).backward()
    assert rel_err(m2.tpg.rnn[1].embedding.weight.grad, g_ref) < 1e-6
    calls = []
    for use_graph in (False, True):
        m = build()
        hook = m.tpg.register_forward_hook(lambda *a: calls.append(1))
        tr = Trainer(m, use_graph=use_graph, warmup_eager=2, recipe=TssimRecipe(5.0, seed=4))
        del calls[:]
        got = float(tr.step(x, None, hr))
        assert len(calls) == 1, calls                            # one student pass per step (the second forward reuses the prior)
        hook.remove()
        assert abs(got - want) < 2e-6 * abs(want), (use_graph, got, want)
        if not use_graph:
            assert rel_err(m.tpg.rnn[1].embedding.weight.grad, g_ref) < 1e-4
            assert rel_err(m.sr.block2.conv1.weight.grad, g_sr_ref) < 1e-4
        ls = [float(tr.step(x, None, hr)) for _ in range(4)]
        assert all(l == l for l in ls)
    # a generator without a prior under the recipe
    torch.manual_seed(1)
    t = tatt_amd.TSRN(**kw).to(dev).train()
    tr = Trainer(t, use_graph=False, recipe=TssimRecipe(5.0, seed=2))
    l0 = float(tr.step(x, None, hr))
    assert l0 == l0 and l0 > 0.0
    # and the Trainer leaves the model uncut: a plain forward/backward reaches every stage's parameters
    out = t(x)
    out.sum().backward()
    assert t.block1[0].weight.grad is not None and float(t.block1[0].weight.grad.abs().max()) > 0.0
