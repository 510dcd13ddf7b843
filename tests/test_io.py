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


def test_collate_labels_follow_the_reference_layout():
    """dataset/dataset.py:2009-2063 (alignCollate_realWTLAMask.__call__): lower-case, spread over 26 steps with
    int((26 - len) / (len - 1)) blanks between characters, characters outside the alphabet dropped, one-hot (B, 37, 1, 26)."""
    from tatt_amd import io
    assert io.stretch_label("ab") == "a" + "-" * 24 + "b"
    assert io.stretch_label("Hello") == "h-----e-----l-----l-----o"
    assert io.stretch_label("x") == "x" and io.stretch_label("") == ""
    assert io.stretch_label("abcdefghijklmnopqrstuvwxyz0123") == "abcdefghijklmnopqrstuvwxyz"
    vecs, masks, tics = io.collate_labels(["Hello", "", "a!b", "zz9"])
    assert tuple(vecs.shape) == (4, 37, 1, 26) and vecs.dtype == torch.float32
    d2a = "-" + io.ALPHABET
    dec = lambda b, n: "".join(d2a[int(vecs[b, :, 0, t].argmax())] for t in range(n))
    assert dec(0, 25) == "h-----e-----l-----l-----o" and float(vecs[0, :, 0, 25].sum()) == 0.0
    assert float(vecs[1].sum()) == 1.0 and float(vecs[1, 0, 0, 0]) == 1.0          # empty word: one blank, tic 0
    # "a!b" -> "a" + 11 blanks + "!" + 11 blanks + "b" (25 steps), the "!" is dropped AFTER stretching: 24 one-hot columns
    assert dec(2, 24) == "a" + "-" * 22 + "b" and float(vecs[2].sum()) == 24.0
    assert tics.tolist() == [1, 0, 1, 1]
    assert masks.numel() == 25 + 1 + 24 + 25 and masks.dtype == torch.int64
    batch = io.collate_batch([(torch.rand(4, 32, 128), torch.rand(4, 16, 64), torch.rand(1, 32, 128), torch.rand(1, 16, 64), w)
                              for w in ("ab", "text")])
    assert len(batch) == 9 and batch[1] is None and tuple(batch[0].shape) == (2, 4, 32, 128) and tuple(batch[2].shape) == (2, 4, 16, 64)
    assert batch[5] == ("ab", "text") and tuple(batch[6].shape) == (2, 37, 1, 26)


def test_ctc_greedy_decode_and_filter():
    """utils/metrics.py:71-92 (get_string_crnn) and utils/util.py:12-32 (str_filt)."""
    from tatt_amd import io
    T, B = 7, 3
    lg = torch.full((T, B, 37), -5.0)
    seqs = [[11, 11, 0, 11, 12, 12, 0], [0, 0, 0, 0, 0, 0, 0], [1, 0, 1, 36, 36, 0, 36]]
    for b, sq in enumerate(seqs):
        for t, c in enumerate(sq):
            lg[t, b, c] = 3.0
    assert io.ctc_greedy_decode(lg) == ["aab", "", "00zz"]
    assert io.str_filt("Ab-C!9", "lower") == "abc9" and io.str_filt("Ab-C!9", "upper") == "AbC9" and io.str_filt("a1", "digit") == "1"


@pytest.mark.gpu
def test_evaluate_reports_recognition_accuracy(dev):
    """The eval loop with a CRNN recogniser: accuracy of SR / LR / HR = fraction of images whose greedy-decoded, filtered string
    equals the filtered label (reference interfaces/super_resolution.py:1527-1558,1662-1664).  Expected strings come from the CPU ORACLE
    of the recogniser (oracle/crnn_oracle.py: own bicubic input, conv / LSTM loops) decoded by a CTC decoder written out here,
    independently of tatt_amd.io -- not from the HIP recogniser's own output."""
    import tatt_amd
    from oracle import crnn_oracle as C
    from tatt_amd import io
    torch.manual_seed(3)
    m = tatt_amd.TSRN(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=2, hidden_units=32).to(dev).eval()
    rec = tatt_amd.CRNN(32, 1, 37, 256)
    sd = {k: v.detach().clone() for k, v in rec.state_dict().items()}
    rec = rec.to(dev).eval()
    g = torch.Generator().manual_seed(0)
    lr, hr = torch.rand(4, 4, 16, 64, generator=g), torch.rand(4, 4, 32, 128, generator=g)

    def oracle_strings(img):
        with torch.no_grad():
            logits = C.crnn_forward(sd, C.parse_crnn_data(img[:, :3]), training=False)          # (T, B, 37)
        out = []
        for b in range(logits.shape[1]):
            best = logits[:, b].argmax(1).tolist()
            chars, prev = [], 0
            for c in best:                                       # CTC greedy: collapse repeats, drop blanks (class 0)
                if c != prev and c != 0:
                    chars.append(("-" + io.ALPHABET)[c])
                prev = c
            out.append("".join(chars))
        return out
    hr_s, lr_s = oracle_strings(hr), oracle_strings(lr)
    labels = [hr_s[0], hr_s[1].upper(), hr_s[2] + "q", lr_s[3] + "!"]
    want_hr = sum(a == b for a, b in zip(hr_s, [io.str_filt(l, "lower") for l in labels])) / 4
    want_lr = sum(a == b for a, b in zip(lr_s, [io.str_filt(l, "lower") for l in labels])) / 4
    res = io.evaluate(m, [(lr.to(dev), hr.to(dev), None, labels)], recognizer=rec)
    assert res["n_images"] == 4
    assert res["accuracy_hr"] == want_hr and res["accuracy_lr"] == want_lr, (res, want_hr, want_lr, hr_s, lr_s)
    assert want_hr >= 0.5 and 0.0 <= res["accuracy"] <= 1.0 and res["psnr"] > 0.0
    assert not rec.training                                       # (it was in eval mode before: restored to that)


def test_collate_images_and_labels_match_the_reference_fixture():
    """tests/golden/collate.npz was produced by the reference's own `alignCollate_realWTLAMask(imgH=32, imgW=128, down_sample_scale=2,
    mask=True)` (tools/gen_golden_collate.py): eight synthetic RGB images of assorted sizes and eight labels (empty, one character,
    longer than 26, characters outside the alphabet).  The PIL resize + ToTensor + mask plane and every label tensor must match."""
    import numpy as np
    from PIL import Image
    from tatt_amd import io
    z = np.load("tests/golden/collate.npz")
    n = int(z["n"])
    alphabet = str(z["alphabet"])
    samples = []
    for i in range(n):
        hr, lr = Image.fromarray(z["hr%d" % i], "RGB"), Image.fromarray(z["lr%d" % i], "RGB")
        samples.append((hr, lr, hr, lr, io.str_filt(str(z["labels_in"][i]), "lower")))
    out = io.collate_pil_batch(samples, imgH=32, imgW=128, down_sample_scale=2, mask=True, alphabet=alphabet)
    images_HR, pseudo, images_lr, images_HRy, images_lry, label_strs, label_vecs, wmask, wtics = out
    assert pseudo is None and list(label_strs) == [str(v) for v in z["label_strs"]]
    assert tuple(images_HR.shape) == (n, 4, 32, 128) and tuple(images_lr.shape) == (n, 4, 16, 64)
    assert torch.equal(images_HR, torch.from_numpy(z["images_HR"])) and torch.equal(images_lr, torch.from_numpy(z["images_lr"]))
    assert set(images_HR[:, 3].unique().tolist()) <= {0.0, 1.0}                      # the mask plane is binary
    assert torch.equal(label_vecs, torch.from_numpy(z["label_vecs"]))
    assert torch.equal(wmask, torch.from_numpy(z["weighted_mask"])) and torch.equal(wtics, torch.from_numpy(z["weighted_tics"]))


def test_lmdb_record_reader_on_an_in_memory_environment():
    """`LmdbRecords` against the reference's record layout (dataset/dataset.py:565-686) with a dict standing in for the lmdb transaction:
    1-based keys, PNG-encoded images decoded to RGB, labels filtered by voc_type, a missing label read as a blank, a broken record
    skipped, and the batch it feeds to `collate_pil_batch`."""
    import io as pyio
    import numpy as np
    from PIL import Image
    from tatt_amd import io
    rng = np.random.default_rng(1)

    def png(w, h):
        b = pyio.BytesIO()
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8), "RGB").save(b, format="PNG")
        return b.getvalue()
    long_label = "x" * 150                                           # longer than max_len = 100
    db = {b"num-samples": b"5"}
    for i, (lab, (w, h)) in enumerate(zip(("Ab-C!9", None, "second", long_label, "fifth"),
                                          ((100, 30), (64, 20), (130, 40), (90, 28), (80, 24))), start=1):
        db[b"image_hr-%09d" % i] = png(w, h)
        db[b"image_lr-%09d" % i] = png(w // 2, h // 2)
        if lab is not None:
            db[b"label-%09d" % i] = lab.encode()
    db[b"image_lr-%09d" % 2] = b"not an image"                     # record 2 (item 1) is unreadable
    rec = io.LmdbRecords(db, voc_type="lower")
    assert len(rec) == 5
    hr, lr, hry, lry, lab = rec[0]
    assert lab == "abc9" and hr.size == (100, 30) and lr.size == (50, 15) and hr.mode == "RGB" and hry.size == hr.size
    # the reference's `return self[index + 1]` after `index += 1`: item 1 falls through to item 3 (record 4), NOT to item 2
    assert rec[1][0].size == (90, 28)
    # ... and its `except IOError or len(word) > self.max_len` catches IOError only: the over-long label of record 4 is returned
    assert rec[3][4] == long_label and rec[1][4] == long_label
    assert rec[2][4] == "second" and rec[2][0].size == (130, 40)
    db[b"image_hr-%09d" % 4] = b"broken"                           # item 3 unreadable: falls through to item 5, past the end
    with pytest.raises(IndexError):
        rec[3]
    batch = io.collate_pil_batch([rec[0], rec[2]])
    assert tuple(batch[0].shape) == (2, 4, 32, 128) and tuple(batch[2].shape) == (2, 4, 16, 64) and batch[5] == ("abc9", "second")
    y = io.rgb_to_yuv_u8(np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0]]], np.uint8))
    assert y[0, 0].tolist() == [255, 128, 128] and y[0, 1].tolist() == [0, 128, 128] and y[0, 2, 0] == 76
    with pytest.raises(ImportError):
        io.open_lmdb("/nonexistent")
