"""CPU: the oracle (oracle/tatt_oracle.py) against the golden vectors generated from the real reference
(tools/gen_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP path with it."""
import numpy as np
import pytest
import torch

from oracle import tatt_oracle as O
from oracle.fixtures import randomize_state_dict, summarize
from tests.util import max_err, rel_err

STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)


def product_sd(cls="TSRN_TL_TRANS", randomize=True, **kw):
    """Seed-1234 state_dict from the product module's constructor (bit-identical to the reference's init,
    see test_module_surface.py) -- so no 30 MB weight file has to be committed."""
    import tatt_amd
    torch.manual_seed(1234)
    sd = getattr(tatt_amd, cls)(**(kw or STD)).state_dict()
    return randomize_state_dict(sd) if randomize else sd


def test_known_answer_vector():
    z = np.load("tests/golden/kat.npz")
    sd = product_sd(randomize=False)
    with torch.no_grad():
        o = O.generator_forward(sd, torch.from_numpy(z["x"]), torch.from_numpy(z["tp"]), training=False)
    assert abs(float(o["sr"].double().sum()) - 168.209915) < 1e-3
    assert max_err(o["sr"], torch.from_numpy(z["sr"])) < 5e-6
    assert max_err(o["pr_weights"], torch.from_numpy(z["pr_weights"])) < 1e-6


@pytest.mark.parametrize("name,cls,tatt,kw", [
    ("tatt_eval_b2", "TSRN_TL_TRANS", True, STD), ("tsrn_eval_b2", "TSRN", False, STD),
    ("large_tile", "TSRN_TL_TRANS", True, dict(scale_factor=2, width=256, height=64, STN=False, mask=True,
                                               srb_nums=5, hidden_units=32))])
def test_eval_forward(name, cls, tatt, kw):
    z = np.load("tests/golden/%s.npz" % name)
    sd = product_sd(cls, **kw)
    with torch.no_grad():
        o = O.generator_forward(sd, torch.from_numpy(z["x"]), torch.from_numpy(z["tp"]) if tatt else None,
                                training=False, tatt=tatt, stn=kw["STN"])
    assert max_err(o["sr"], torch.from_numpy(z["sr"])) < 5e-6
    assert max_err(o["block1"][:, :8], torch.from_numpy(z["block1"])) < 1e-5
    assert max_err(o["block7"][:, :8], torch.from_numpy(z["block7"])) < 1e-5
    if tatt:
        assert max_err(o["pr_weights"], torch.from_numpy(z["pr_weights"])) < 1e-6


@pytest.mark.parametrize("name,cls,tatt", [("tatt_train_b4", "TSRN_TL_TRANS", True), ("tsrn_train_b3", "TSRN", False)])
def test_train_step(name, cls, tatt):
    z = np.load("tests/golden/%s.npz" % name)
    sd = product_sd(cls)
    x, hr = torch.from_numpy(z["x"]), torch.from_numpy(z["hr"])
    tp = torch.from_numpy(z["tp"]) if tatt else None
    loss, grads, sd1, _, out, total = O.train_step(sd, x, tp, hr, tatt=tatt, stn=True)
    assert abs(float(loss) - float(z["loss"])) < 1e-4 * float(z["loss"])
    assert max_err(out["sr"], torch.from_numpy(z["sr"])) < 3e-4          # conditioning: see tests/golden/REPORT.txt
    assert abs(float(total) - float(z["gnorm"])) < 1e-3 * float(z["gnorm"])
    assert sorted(k for k, g in grads.items() if g is None) == sorted(z["none_keys"].tolist())
    noise = set(z["noise_keys"].tolist())
    for k, ref in zip(z["grad_keys"].tolist(), z["grad_summary"]):
        if k in noise:
            continue
        got = summarize(grads[k])
        assert abs(got[0] - ref[0]) < 1e-2 * ref[0] + 1e-7, (k, got[0], ref[0])
    for key in z.files:
        if key.startswith("g:"):
            assert rel_err(grads[key[2:]], torch.from_numpy(z[key])) < 1e-2, key
    for k, ref in zip(z["w1_keys"].tolist(), z["w1_summary"]):
        if k in noise:
            continue
        got = summarize(sd1[k].float())
        assert abs(got[0] - ref[0]) < 1e-3 * abs(ref[0]) + 1e-5, k          # post-Adam weight norms


def test_query_gru_batch_axis_quirk():
    z = np.load("tests/golden/qgru.npz")
    sd = product_sd(scale_factor=2, width=128, height=32, STN=False)
    for B in (1, 2, 4):
        q = O.query_embedding(sd, "infoGen", B, 16, 64)
        assert max_err(q[:, ::37], torch.from_numpy(z["q%d" % B])) < 1e-5
    # the quirk itself: a sample's embedding depends on its index in the batch and on B
    q4 = O.query_embedding(sd, "infoGen", 4, 16, 64)
    assert max_err(q4[0], q4[1]) > 1e-4


def test_tps_out_of_range_control_points():
    z = np.load("tests/golden/tps.npz")
    sd = product_sd()
    y, src = O.tps_transform(torch.from_numpy(z["x"]), torch.from_numpy(z["ctrl"]), sd, "tps")
    assert float(src.min()) < 0 and float(src.max()) > 1            # the clamp is exercised
    assert max_err(y, torch.from_numpy(z["y"])) < 1e-5
    assert max_err(src, torch.from_numpy(z["src"])) < 1e-5


TBSRN_KW = dict(scale_factor=2, width=512, height=32, STN=True, mask=True, input_channel=4)


def test_tbsrn_forward_and_gradients():
    """TBSRN variant at LR 16x256 -- the only size the unmodified reference executes."""
    z = np.load("tests/golden/tbsrn_b2.npz")
    sd = product_sd("TBSRN", **TBSRN_KW)
    assert list(sd.keys()) == z["sd_keys"].tolist()
    x, hr = torch.from_numpy(z["x"]), torch.from_numpy(z["hr"])
    with torch.no_grad():
        o = O.tbsrn_forward(sd, x, training=False)
    assert max_err(o["sr"], torch.from_numpy(z["sr_eval"])) < 2e-5
    loss, grads, _, _, out, _ = O.train_step(sd, x, None, hr, stn=False, tbsrn=True)
    assert max_err(out["sr"], torch.from_numpy(z["sr_train"])) < 2e-5
    assert abs(float(loss) - float(z["loss"])) < 1e-4 * float(z["loss"])
    assert sorted(k for k, g in grads.items() if g is None) == sorted(z["none_keys"].tolist())
    scale = max(float(r[0]) for r in z["grad_summary"])          # largest gradient l2 norm
    for k, ref in zip(z["grad_keys"].tolist(), z["grad_summary"]):
        got = summarize(grads[k])
        assert abs(got[0] - ref[0]) < 1e-2 * ref[0] + 1e-6 * scale * grads[k].numel() ** 0.5, (k, got[0], ref[0])


def test_tbsrn_layer_norm_is_not_nn_layernorm():
    """model/tbsrn.py:23-36: unbiased std with eps outside the root -- differs from nn.LayerNorm by ~1/(2C)."""
    x = torch.randn(5, 128)
    a, b = torch.rand(128) + 0.5, torch.randn(128)
    mine = O.tbsrn_layer_norm(x, a, b)
    ref = a * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + 1e-6) + b
    assert max_err(mine, ref) < 1e-5
    assert max_err(mine, O.layer_norm(x, a, b, 1e-6)) > 1e-3


def test_semantic_loss_and_psnr():
    z = np.load("tests/golden/losses.npz")
    pred = torch.from_numpy(z["pred"]).requires_grad_(True)
    loss = O.semantic_loss(pred, torch.from_numpy(z["gt"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(z["sem"])) < 1e-6
    assert max_err(pred.grad, torch.from_numpy(z["dpred"])) < 1e-8
    assert abs(float(O.calculate_psnr(torch.from_numpy(z["a"]), torch.from_numpy(z["b"]))) - float(z["psnr"])) < 1e-4


def test_ssim_tri_ssim_and_rotation():
    """SURVEY.md 8f-2 (oracle side; kernels are next round): SSIM / TRI_SSIM and torch_distortion against reference vectors."""
    z = np.load("tests/golden/losses.npz")
    a, b, c = (torch.from_numpy(z[k]) for k in ("a", "b", "c"))
    assert abs(float(O.ssim(a, b)) - float(z["ssim"])) < 1e-6
    assert max_err(O.ssim(a, b, size_average=False), torch.from_numpy(z["ssim_per_sample"])) < 1e-6
    assert abs(float(O.tri_ssim(a, b, c)) - float(z["tri_ssim"])) < 1e-6
    d = O.torch_distortion(a, torch.from_numpy(z["arcs"]), torch.from_numpy(z["offs"]))
    assert max_err(d[:, :, ::4, ::4], torch.from_numpy(z["distorted"])) < 1e-5


LARGE = dict(scale_factor=2, width=256, height=64, STN=False, mask=True, srb_nums=5, hidden_units=32)


def test_large_tile_train_step():
    """BASELINE.json configs[4] geometry (LR 32x128), train-mode forward + backward, B=2: oracle vs the reference's vectors."""
    z = np.load("tests/golden/large_train_b2.npz")
    sd = product_sd("TSRN_TL_TRANS", **LARGE)
    x, tp, hr = (torch.from_numpy(z[k]) for k in ("x", "tp", "hr"))
    loss, grads, sd1, _, out, _ = O.train_step(sd, x, tp, hr, tatt=True, stn=False)
    assert abs(float(loss) - float(z["loss"])) < 1e-5 * float(z["loss"])
    assert max_err(out["sr"], torch.from_numpy(z["sr"])) < 2e-5
    assert sorted(k for k, g in grads.items() if g is None) == sorted(z["none_keys"].tolist())
    for key in z.files:
        if key.startswith("g:"):
            assert rel_err(grads[key[2:]], torch.from_numpy(z[key])) < 1e-3, key
    assert max_err(sd1["block4.bn1.running_var"], torch.from_numpy(z["bn_var"])) < 1e-6


def test_bench_first_step_losses():
    """bench.py's known-answer check: the oracle reproduces the reference's first-step loss on bench.py's own model and batch."""
    import json
    from bench import TILES
    want = json.load(open("tests/golden/bench_losses.json"))
    for tile in ("std", "large"):
        t = TILES[tile]
        sd = product_sd("TSRN_TL_TRANS", randomize=False, scale_factor=2, width=2 * t["W"], height=2 * t["H"], STN=t["stn"], mask=True,
                        srb_nums=5, hidden_units=32)
        g = torch.Generator().manual_seed(0)
        B = t["batch"]
        x = torch.rand(B, 4, t["H"], t["W"], generator=g)
        x[:, 3] = (x[:, 3] > 0.5).float()
        hr = torch.rand(B, 4, 2 * t["H"], 2 * t["W"], generator=g)
        tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1)
        with torch.no_grad():
            o = O.generator_forward(sd, x, tp, training=True, tatt=True, stn=t["stn"])
            loss = float(O.image_loss(o["sr"], hr).mean() * 100)
        assert abs(loss - want[t["loss_key"]]) < 1e-4 * loss, (tile, loss, want[t["loss_key"]])


def test_fp64_error_bars_reproduce():
    """The fp64 yardstick: the oracle in float64 reproduces the stored fp64 loss, and its own fp32 errors match the stored ones."""
    z = np.load("tests/golden/tatt_train_b4.npz")
    bars = np.load("tests/golden/fp64_error_bars.npz")
    sd = product_sd("TSRN_TL_TRANS")
    x, hr, tp = (torch.from_numpy(z[k]) for k in ("x", "hr", "tp"))
    l64, g64, _, _, _, _ = O.train_step_fp64(sd, x, tp, hr, tatt=True, stn=True)
    assert abs(float(l64) - float(bars["loss64"])) < 1e-10 * float(l64)
    assert g64["block2.conv1.weight"].dtype == torch.float64
    med = float(np.median(bars["ref32_err"]))
    assert 1e-6 < med < 1e-3                       # the reference's own fp32 gradients: ~2e-5 from fp64 in the median, 7e-3 in the STN head
