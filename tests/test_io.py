"""checkpoint files in the reference's format (interfaces/base.py:621-672 write, :398-443 resume) and the metric
part of the eval loop.  The file round trip needs no GPU; the eval loop runs the HIP path."""
import os

import pytest
import torch

from oracle import tatt_oracle as O
from oracle.fixtures import randomize_state_dict, make_inputs

STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)


def _model(seed=1234):
    import tatt_amd
    torch.manual_seed(seed)
    m = tatt_amd.TSRN_TL_TRANS(**STD)
    m.load_state_dict(randomize_state_dict(m.state_dict()))
    return m


def test_checkpoint_round_trip_in_reference_format(tmp_path):
    from tatt_amd.io import save_checkpoint, load_generator
    m = _model()
    files = save_checkpoint([m], 3, 1200, {"easy": 0.5}, {"easy": {"accuracy": 0.5}}, True, [1.0, 0.5], str(tmp_path), arch="tatt")
    assert [os.path.basename(f) for f in files] == ["model_best_acc_0.pth"]
    blob = torch.load(files[0])
    assert set(blob) == {"state_dict_G", "info", "best_history_res", "best_model_info", "param_num", "converge"}
    assert blob["info"] == {"arch": "tatt", "iters": 1200, "epochs": 3, "batch_size": 48, "voc_type": "all", "up_scale_factor": 2}
    assert blob["param_num"] == 7608334 and len(blob["state_dict_G"]) == 304
    # resume from the DIRECTORY (reference: model_best_acc_<iter>.pth, strict=False) and from the FILE (strict)
    for resume in (str(tmp_path), files[0]):
        fresh = _model(seed=7)
        info = load_generator(fresh, resume)
        assert info["iters"] == 1200
        for (k, a), (_, b) in zip(m.state_dict().items(), fresh.state_dict().items()):
            assert torch.equal(a, b), k
    # non-best: checkpoint.pth; a DataParallel checkpoint ('module.' keys) and a bare state_dict load as well
    files = save_checkpoint([m], 3, 1300, {}, {}, False, [], str(tmp_path))
    assert os.path.basename(files[0]) == "checkpoint.pth"
    p = str(tmp_path / "dp.pth")
    torch.save({"state_dict_G": {"module." + k: v for k, v in m.state_dict().items()}}, p)
    fresh = _model(seed=8)
    load_generator(fresh, p)
    assert torch.equal(fresh.state_dict()["block1.0.weight"], m.state_dict()["block1.0.weight"])
    p = str(tmp_path / "bare.pth")
    torch.save(m.state_dict(), p)
    assert load_generator(_model(seed=9), p) is None


@pytest.mark.gpu
def test_eval_loop_metrics(dev):
    from tatt_amd.io import evaluate
    m = _model().to(dev)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    batches, want_p, want_s = [], 0.0, 0.0
    for i in range(2):
        x, tp, hr = make_inputs(3, seed=60 + i)
        batches.append((x.to(dev), hr.to(dev), tp.to(dev)))
        with torch.no_grad():
            sr = O.generator_forward(sd, x, tp, training=False)["sr"]
        want_p += float(O.calculate_psnr(sr, hr)) / 2
        want_s += float(O.ssim(sr, hr)) / 2
    got = evaluate(m, batches)
    assert got["n_batches"] == 2 and m.training
    assert abs(got["psnr"] - want_p) < 1e-3 and abs(got["ssim"] - want_s) < 1e-5, (got, want_p, want_s)
