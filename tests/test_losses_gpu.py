"""SURVEY.md 8f-2 on the GPU: SSIM / TRI_SSIM losses and the rotation augmentation (torch_distortion) of the shipped recipe
(train_TATT.sh: --tssim_loss --rotate_train=5) -- HIP kernels against the reference-generated vectors (tests/golden/losses.npz)
and, with gradients, against the oracle restatement (pinned to the reference in tests/test_oracle_golden.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import tatt_oracle as O
from tests.util import compare_fn, max_err, rel_err

pytestmark = pytest.mark.gpu


def _z():
    return np.load("tests/golden/losses.npz")


def test_ssim_and_tri_ssim_against_reference_vectors(dev):
    from tatt_amd.losses import SSIM, TRI_SSIM
    z = _z()
    a, b, c = (torch.from_numpy(z[k]).to(dev) for k in ("a", "b", "c"))
    assert abs(float(SSIM()(a, b)) - float(z["ssim"])) < 2e-6
    assert abs(float(TRI_SSIM()(a, b, c)) - float(z["tri_ssim"])) < 2e-6
    assert max_err(SSIM(size_average=False)(a, b), torch.from_numpy(z["ssim_per_sample"])) < 2e-6
    # the SR image reaches the loss as an NCHW-shaped view of NHWC memory
    a_cl = a.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert not a_cl.is_contiguous()
    assert abs(float(TRI_SSIM()(a_cl, b, c)) - float(z["tri_ssim"])) < 2e-6


@pytest.mark.parametrize("shape", [(3, 4, 32, 128), (2, 3, 16, 64), (2, 4, 21, 70)])
def test_ssim_gradients(dev, shape):
    from tatt_amd import functional as Fh
    g = torch.Generator().manual_seed(3)
    a, b, c = (torch.rand(shape, generator=g) for _ in range(3))
    b = (0.7 * a + 0.3 * b)                                      # correlated images: the regime the loss lives in
    compare_fn("tri_ssim", lambda x, y, w: Fh.SsimFn.apply(x, y, w), lambda x, y, w: O.tri_ssim(x, y, w, size_average=False),
               [a, b, c], dev, rtol=2e-5, atol=2e-6, grtol=2e-4, gatol=2e-5)
    compare_fn("ssim", lambda x, y: Fh.SsimFn.apply(x, y, None), lambda x, y: O.ssim(x, y, size_average=False),
               [a[:, :3].contiguous(), b[:, :3].contiguous()], dev, rtol=2e-5, atol=2e-6, grtol=2e-4, gatol=2e-5)


def test_rotation_against_reference_vector(dev):
    from tatt_amd.losses import torch_distortion
    z = _z()
    a = torch.from_numpy(z["a"])
    arcs, offs = torch.from_numpy(z["arcs"]), torch.from_numpy(z["offs"])
    y = torch_distortion(a.to(dev), arcs, offs)
    assert tuple(y.shape) == tuple(a.shape)
    assert max_err(y[:, :, ::4, ::4], torch.from_numpy(z["distorted"])) < 2e-5
    assert max_err(y, O.torch_distortion(a, arcs, offs)) < 2e-5


@pytest.mark.parametrize("deg,H,W", [(5.0, 32, 128), (5.0, 16, 64), (25.0, 32, 128), (60.0, 24, 40)])
def test_rotation_image_gradient(dev, deg, H, W):
    """Gradient w.r.t. the IMAGE through the resampler (the recipe rotates a network output): deterministic gather kernel vs
    autograd through the oracle, including strong rotations and the extremes of the aspect jitter."""
    from tatt_amd.losses import torch_distortion
    g = torch.Generator().manual_seed(int(deg))
    x = torch.rand(5, 4, H, W, generator=g)
    arcs = torch.tensor([deg, -deg, 0.3 * deg, 0.0, -0.7 * deg]) / 180.0 * math.pi
    offs = torch.tensor([0.0, 1.0, 0.5, 0.25, 0.9])
    compare_fn("rotate", lambda t: torch_distortion(t, arcs, offs), lambda t: O.torch_distortion(t, arcs, offs), [x], dev,
               rtol=2e-4, atol=2e-5, grtol=2e-4, gatol=2e-5)


def test_tssim_recipe_composition(dev):
    """(1 - TRI_SSIM(rotate(sr_ret), sr, hr).mean()) * 10 as in interfaces/super_resolution.py:910-914: value and both gradients."""
    from tatt_amd.losses import TRI_SSIM, torch_distortion
    g = torch.Generator().manual_seed(11)
    sr, sr_ret, hr = (torch.rand(4, 4, 32, 128, generator=g) for _ in range(3))
    arcs = (torch.rand(4, generator=g) * 2 - 1) * 5 / 180 * math.pi
    offs = torch.rand(4, generator=g)

    def hip(u, v):
        return (1 - TRI_SSIM()(torch_distortion(v, arcs, offs), u, hr.to(dev)).mean()) * 10.

    def ref(u, v):
        return (1 - O.tri_ssim(O.torch_distortion(v, arcs, offs), u, hr)) * 10.
    compare_fn("tssim_recipe", hip, ref, [sr, sr_ret], dev, rtol=2e-5, atol=2e-5, grtol=3e-4, gatol=3e-5)


def test_image_loss_class(dev):
    from tatt_amd.losses import ImageLoss
    g = torch.Generator().manual_seed(2)
    sr, hr = torch.rand(3, 4, 32, 128, generator=g), torch.rand(3, 4, 32, 128, generator=g)
    got = ImageLoss(gradient=True, loss_weight=[1, 1e-4])(sr.to(dev), hr.to(dev))
    assert max_err(got, O.image_loss(sr, hr)) < 1e-6


def test_tssim_recipe_trainer_step(dev):
    """One Trainer step of the shipped recipe (rotation, two generator forwards, ImageLoss + TRI_SSIM) against the oracle
    composition: loss, gradient norm, a few complete gradients (dropout off, STN off for conditioning); then the same step replayed
    as a hipGraph equals the eager step."""
    import tatt_amd
    from oracle.fixtures import randomize_state_dict, make_inputs
    from tatt_amd.train import Trainer, TssimRecipe
    kw = dict(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32)

    def build():
        torch.manual_seed(1234)
        m = tatt_amd.TSRN_TL_TRANS(**kw)
        m.load_state_dict(randomize_state_dict(m.state_dict()))
        m = m.to(dev).train()
        m.infoGen.dropout_on = False
        return m
    m = build()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, tp, hr = make_inputs(3, seed=21)
    rec = TssimRecipe(5.0, seed=4)
    tr = Trainer(m, use_graph=False, recipe=rec)
    loss = tr.step(x.to(dev), tp.to(dev), hr.to(dev))
    arcs, offs = rec.last
    # oracle composition (BatchNorm running statistics do not influence train-mode outputs)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd0.items() if O.is_param(k)}
    full = dict(sd0, **leaves)
    x_rot, hr_rot = O.torch_distortion(x, arcs, offs), O.torch_distortion(hr, arcs, offs)
    x_ret = O.torch_distortion(x_rot, -arcs, offs)
    sr = O.generator_forward(full, x_rot, tp, training=True, tatt=True, stn=False)["sr"]
    sr_ret = O.generator_forward(full, x_ret, tp, training=True, tatt=True, stn=False)["sr"]
    o_loss = O.image_loss(sr, hr_rot).mean() * 100 + (1 - O.tri_ssim(O.torch_distortion(sr_ret, arcs, offs), sr, hr_rot)) * 10
    o_loss.backward()
    assert abs(float(loss) - float(o_loss)) < 2e-5 * abs(float(o_loss)), (float(loss), float(o_loss))
    total = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in leaves.values() if v.grad is not None))
    assert abs(float(tr.last_grad_norm) - float(total)) < 1e-3 * float(total)
    params = dict(m.named_parameters())
    for k in ("block8.1.weight", "block4.conv1.weight", "block2.gru1.gru.weight_hh_l0", "infoGen.fc_in.weight",
              "infoGen.transformer.decoder.layers.1.linear1.weight", "block1.0.weight", "infoGen.init_factor.weight"):
        e = rel_err(params[k].grad, leaves[k].grad)
        assert e < 2e-3, (k, e)
    # hipGraph replay of the recipe: same angles, same result as eager
    def run(use_graph):
        mm = build()
        t = Trainer(mm, use_graph=use_graph, warmup_eager=2, recipe=TssimRecipe(5.0, seed=9))
        ls = [float(t.step(x.to(dev), tp.to(dev), hr.to(dev))) for _ in range(5)]
        return ls, t.flat_p.clone()
    le, pe = run(False)
    lg, pg = run(True)
    assert le == lg and torch.equal(pe, pg)
